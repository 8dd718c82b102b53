#!/bin/bash
# Measurement-only builds of libgnna.so whose sweep kernel fetches the source rows with a cache policy that bypasses the CU's
# vector L1 (buffer loads with aux = sc1 / sc0 sc1; MI355X_MICROARCH.md: L2-served, L1 bypassed) -- does the ~0 %-hit L1 cost
# the gather anything?  Results are exact (only the policy of the row loads changes).  Built from a temporary copy of csrc/.
#   libgnna_rows_sc1.so, libgnna_rows_sc0sc1.so, libgnna_rows_buffer_plain.so (the same buffer-load form, default policy)
# use: GNNA_LIB=tools/ceiling/libgnna_rows_sc1.so python tools/probe_phases.py 128 16
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
cp -r "$ROOT/gnnadvisor_osdi21_amd/csrc" "$TMP/csrc"
python3 - "$TMP/csrc/gnna_sweep.hip" <<'PY'
import sys
p = sys.argv[1]
s = open(p).read()
a = "for (int u = 0; u < U; u++) v[u] = *row_ptr(offs[u * RPI + lslot]);"
b = "v[u] = *row_ptr(nn[u]);"
c = "                        const int nb = (nr + U - 1) / U;"
assert s.count(a) == 1 and s.count(b) == 1 and s.count(c) == 1
s = s.replace(a, "for (int u = 0; u < U; u++) v[u] = load_row(offs[u * RPI + lslot]);")
s = s.replace(b, "v[u] = load_row(nn[u]);")
s = s.replace(c, """                        auto load_row = [&](uint32_t o) -> VT {
                            typedef int i32x4_t __attribute__((ext_vector_type(4)));
                            const i32x4_t r = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, (int)(o + col_off), 0, GNNA_ROW_AUX);
                            return __builtin_bit_cast(VT, r);
                        };
""" + c)
d = "    const char *xbase = reinterpret_cast<const char *>(p.X);"
assert s.count(d) == 1
s = s.replace(d, d + "\n    __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.X), (short)0, 0x7fffffff, 0x00020000);")
open(p, "w").write(s)
PY
cd "$TMP/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -fvisibility=hidden -I$ROOT/include -I."
build() {   # name, aux
  local objs=$(ls "$ROOT"/gnnadvisor_osdi21_amd/csrc/build/*.o | grep -v gnna_sweep.hip)
  /opt/rocm/bin/hipcc $FLAGS -DGNNA_ROW_AUX=$2 -c gnna_sweep.hip -o "$TMP/$1.o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden $objs "$TMP/$1.o" -o "$HERE/libgnna_$1.so"
}
build rows_buffer_plain 0 &
build rows_sc1 16 &
build rows_sc0sc1 17 &
wait
ls -la "$HERE"/libgnna_rows_*.so
