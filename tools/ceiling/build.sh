#!/bin/bash
# builds the measurement-only gather floor kernel next to this script
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared gather_ceiling.hip -o libceiling.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 lds_accumulate_bench.hip -o lds_accumulate_bench
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared push_probe.hip -o libpush.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared push_probe2.hip -o libpush2.so
for r in 72 64; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DPUSH_R=$r push_probe3.hip -o libpush3_r$r.so; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared push_probe4.hip -o libpush4.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared l1_policy_probe.hip -o libl1policy.so
