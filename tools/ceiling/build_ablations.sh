#!/bin/bash
# Timing-only builds of libgnna.so with parts of a kernel compiled out (WRONG RESULTS; measurement only).  The switches are
# NOT in the product sources (VERDICT r4 task 8): ablation_switches.patch adds them to a temporary copy of csrc/, which is
# what gets compiled here.
#   libgnna_nofold.so / _nolocks.so / _nofold_nolocks.so -- sweep kernel: rows never folded / accumulated in LDS, no chunk locks
#   libgnna_sddmm_noa.so / _noout.so / _none.so          -- SDDMM path: destination row never fetched, edge_out never written
# use: GNNA_LIB=tools/ceiling/libgnna_nofold.so python bench.py --headline-only
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
mkdir -p "$TMP/gnnadvisor_osdi21_amd"
cp -r "$ROOT/gnnadvisor_osdi21_amd/csrc" "$TMP/gnnadvisor_osdi21_amd/csrc"
(cd "$TMP" && patch -p1 -s < "$HERE/ablation_switches.patch")
cd "$TMP/gnnadvisor_osdi21_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -fvisibility=hidden -I$ROOT/include -I."
build() {   # name, source, defines
  local objs=$(ls "$ROOT"/gnnadvisor_osdi21_amd/csrc/build/*.o | grep -v "$2")
  /opt/rocm/bin/hipcc $FLAGS $3 -c $2 -o "$TMP/$1.o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden $objs "$TMP/$1.o" -o "$HERE/libgnna_$1.so"
}
build nofold gnna_sweep.hip "-DGNNA_ABLATE_FOLD"
build nolocks gnna_sweep.hip "-DGNNA_ABLATE_LOCKS"
build nofold_nolocks gnna_sweep.hip "-DGNNA_ABLATE_FOLD -DGNNA_ABLATE_LOCKS"
build sddmm_noa gnna_stream.hip "-DGNNA_SDDMM_ABLATE_A"
build sddmm_noout gnna_stream.hip "-DGNNA_SDDMM_ABLATE_OUT"
build sddmm_none gnna_stream.hip "-DGNNA_SDDMM_ABLATE_A -DGNNA_SDDMM_ABLATE_OUT"
ls -la "$HERE"/libgnna_*.so
