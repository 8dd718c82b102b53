#!/bin/bash
# Timing-only builds of libgnna.so with parts of the sweep kernel compiled out (WRONG RESULTS; measurement only):
#   libgnna_nofold.so  -- rows are never folded / accumulated in LDS          (-DGNNA_ABLATE_FOLD)
#   libgnna_nolocks.so -- no chunk locks                                         (-DGNNA_ABLATE_LOCKS)
# use: GNNA_LIB=tools/ceiling/libgnna_nofold.so python bench.py --headline-only
set -e
cd "$(dirname "$0")/../../gnnadvisor_osdi21_amd/csrc"
OUT=../../tools/ceiling
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -fvisibility=hidden -I../../include -I."
OBJS=$(ls build/*.o | grep -v gnna_sweep)
for v in "nofold:-DGNNA_ABLATE_FOLD" "nolocks:-DGNNA_ABLATE_LOCKS" "nofold_nolocks:-DGNNA_ABLATE_FOLD -DGNNA_ABLATE_LOCKS"; do
  name=${v%%:*}; defs=${v#*:}
  /opt/rocm/bin/hipcc $FLAGS $defs -c gnna_sweep.hip -o /tmp/gnna_sweep_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden $OBJS /tmp/gnna_sweep_$name.o -o $OUT/libgnna_$name.so
done
ls -la $OUT/libgnna_*.so
