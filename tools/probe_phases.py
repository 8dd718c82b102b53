"""Kernel time of the prepared Reddit-like headline (D = 64) over neighbor-group sizes x forced column-phase counts
(finer slices than the rule's 16: do 1.9-2.5 MB slices, which leave room in an XCD's 4 MiB L2, pay for their smaller work
items?).    python tools/probe_phases.py [partSizes] [phases]      e.g. 128,192,256 12,16,20,24,32"""
import gc
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from gnnadvisor_osdi21_amd import _lib

parts = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "128,192,256").split(",")]
phases = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "12,16,20,24,32").split(",")]
dev = torch.device("cuda", 0)
_lib.load()
for ps in parts:
    for b in phases:
        w = bench.Workload("reddit-like", 64, dev, scale=1.0, locality=0.0, manual=False, part_size=ps, lifecycle="prepared",
                           force_phases=b)
        e, p = w.time(20, 5, blocks=3)
        chk = w.verify(16)
        print(f"partSize {ps:4d} phases {b:3d} (used {w.phases:3d}, swept {w.swept}): kernel {p['main_ms']:.4f} ms  step "
              f"{min(w.block_ms):.4f} ms  verified {chk['verified']}  tuning slack {_lib.get_tuning().get('sweep_slack')}", flush=True)
        _lib.release_graph(w.g.column_index)
        del w
        _lib.reset_tuning()
        gc.collect()
        torch.cuda.empty_cache()
