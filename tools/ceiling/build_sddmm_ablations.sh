#!/bin/bash
# Timing-only builds of libgnna.so with parts of the SDDMM path of stream_kernel compiled out (WRONG RESULTS; measurement only):
#   libgnna_sddmm_noa.so      -- destination row never fetched nor permuted   (-DGNNA_SDDMM_ABLATE_A)
#   libgnna_sddmm_noout.so    -- edge_out never written                        (-DGNNA_SDDMM_ABLATE_OUT)
# use: GNNA_LIB=tools/ceiling/libgnna_sddmm_noa.so python tools/probe_sddmm.py reddit-like 64 64
set -e
cd "$(dirname "$0")/../../gnnadvisor_osdi21_amd/csrc"
OUT=../../tools/ceiling
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -fvisibility=hidden -I../../include -I."
OBJS=$(ls build/*.o | grep -v gnna_stream)
for v in "noa:-DGNNA_SDDMM_ABLATE_A" "noout:-DGNNA_SDDMM_ABLATE_OUT" "none:-DGNNA_SDDMM_ABLATE_A -DGNNA_SDDMM_ABLATE_OUT"; do
  name=${v%%:*}; defs=${v#*:}
  /opt/rocm/bin/hipcc $FLAGS $defs -c gnna_stream.hip -o /tmp/gnna_stream_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden $OBJS /tmp/gnna_stream_$name.o -o $OUT/libgnna_sddmm_$name.so
done
ls -la $OUT/libgnna_sddmm_*.so
