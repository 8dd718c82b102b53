#!/bin/bash
# Register / spill / LDS / occupancy table of the kernels of one source of libgnna.so (the compiler's own
# kernel-resource-usage remarks; no GPU needed).   usage: tools/resource_usage.sh gnna_sweep.hip [extra hipcc flags]
cd "$(dirname "$0")/../gnnadvisor_osdi21_amd/csrc" || exit 1
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -fvisibility=hidden -I../../include -I. "$@" -c "$f" -o /tmp/ru.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re,subprocess
cur=None
rows=[]
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)',line)
    if m:
        cur={'name':m.group(1)}; rows.append(cur); continue
    m=re.search(r'remark:\s+(\w[\w \[\]/]*): (\S+)',line)
    if m and cur is not None: cur[m.group(1).strip()]=m.group(2)
for r in rows:
    n=subprocess.run(['c++filt',r['name']],capture_output=True,text=True).stdout.strip()
    n=re.sub(r'gnna::\(anonymous namespace\)::','',n); n=re.sub(r'\(.*','',n); n=re.sub(r'^void ','',n)
    g=lambda k: r.get(k,'?')
    print(f\"{n:52s} SGPR {g('TotalSGPRs'):>4s} VGPR {g('VGPRs'):>4s} AGPR {g('AGPRs'):>3s} sgpr-spill {g('SGPRs Spill'):>3s} vgpr-spill {g('VGPRs Spill'):>3s} scratch {g('ScratchSize [bytes/lane]'):>4s} waves/SIMD {g('Occupancy [waves/SIMD]'):>2s} LDS {g('LDS Size [bytes/block]')}\")
"
