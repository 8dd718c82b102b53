#!/bin/bash
# Round 6, timing only (WRONG RESULTS): SDDMM on stream_kernel with parts compiled out of a temporary copy of the source.
#   libgnna_sddmm_noreduce.so  -- the 16-lane fold of a slot's partial dot product is skipped
#   libgnna_sddmm_nopark.so    -- the dot is neither parked in LDS nor written out
#   libgnna_sddmm_noafetch.so  -- the destination piece is never fetched / permuted
# use: GNNA_LIB=tools/ceiling/libgnna_sddmm_noreduce.so python tools/probe_sddmm_ablate.py
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"; ROOT="$(cd "$HERE/../.." && pwd)"
TMP=$(mktemp -d); trap 'rm -rf "$TMP"' EXIT
SRC=$ROOT/gnnadvisor_osdi21_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -fvisibility=hidden -I$ROOT/include -I$SRC"
build() {   # name, sed expression
  sed -e "$2" $SRC/gnna_stream.hip > $TMP/gnna_stream_$1.hip
  if cmp -s $SRC/gnna_stream.hip $TMP/gnna_stream_$1.hip; then echo "ablation $1 did not change the source"; exit 1; fi
  /opt/rocm/bin/hipcc $FLAGS -c $TMP/gnna_stream_$1.hip -o $TMP/$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden $(ls $SRC/build/*.o | grep -v gnna_stream) $TMP/$1.o -o $HERE/libgnna_sddmm_$1.so
}
build noreduce 's/float dot = lane_group_sum<LPR>((prod\[0\] + prod\[1\]) + (prod\[2\] + prod\[3\]));/float dot = (prod[0] + prod[1]) + (prod[2] + prod[3]);/'
build nopark 's/if (c == 0) pend\[j \* RPI + slot\] = dot;/if (dot == 12345.678f) pend[j * RPI + slot] = dot;/'
build noafetch 's/if (fresh(u)) a\[u\] = a_load(u);/if (false) a[u] = a_load(u);/; s/if (fresh(jn + u)) a\[u\] = a_load(jn + u);/if (false) a[u] = a_load(jn + u);/'
ls -la $HERE/libgnna_sddmm_*.so
