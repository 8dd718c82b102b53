"""Floor of the sliced gather (equal id ranges, as today) when the COLD rows of every slice are loaded non-temporally:
per slice the `keep_mb` MB of most frequently gathered rows are loaded normally, every other row with an `nt` load (first
to leave the L2), so that the L2 keeps what is worth keeping instead of what came last.
usage: probe_cold_nt.py [config]   (build first: tools/ceiling/build.sh)"""
import ctypes
import json
import os
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
from gnnadvisor_osdi21_amd import graph  # noqa: E402

lib = ctypes.CDLL(os.path.join(here, "libceiling.so"))
lib.gather_ceiling_nt_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
cfg = sys.argv[1] if len(sys.argv) > 1 else "reddit-like"
D = 64
g = graph.make_config_graph(cfg, device=dev)
N, nnz = g.num_nodes, g.column_index.numel()
X = torch.randn(N, D, device=dev)
col = g.column_index
out = torch.empty((nnz // 256 + 64) * 256, device=dev)
freq = torch.bincount(col.long(), minlength=N)


def floor_ms(ids, seg=512, n=10):
    def go():
        assert lib.gather_ceiling_nt_launch(X.data_ptr(), ids.data_ptr(), ids.numel(), D, seg, out.data_ptr()) == 0
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        go()
    b.record()
    torch.cuda.synchronize()
    return round(a.elapsed_time(b) / n, 4)


for B in (4, 8, 12, 16):
    phase_of = torch.div(torch.arange(N, device=dev) * B, N, rounding_mode="floor")
    ph = phase_of[col.long()].to(torch.int16)
    order = torch.sort(ph, stable=True).indices
    base_ids = col[order].contiguous()
    row = dict(cfg=cfg, D=D, B=B, all_normal=floor_ms(base_ids), keep={})
    for keep_mb in (1.0, 2.0, 3.0, 3.5, 4.0):
        keep_rows = int(keep_mb * 1e6 / (D * 4))
        cold = torch.ones(N, dtype=torch.bool, device=dev)
        for p in range(B):
            idx = (phase_of == p).nonzero().flatten()
            top = idx[torch.topk(freq[idx], min(keep_rows, idx.numel())).indices]
            cold[top] = False
        marked = torch.where(cold[base_ids.long()], (base_ids.long() | 0x80000000) - (1 << 32), base_ids.long()).to(torch.int32)
        row["keep"][keep_mb] = dict(ms=floor_ms(marked), cold_edge_share=round(float(cold[col.long()].float().mean()), 3))
    print(json.dumps(row), flush=True)
