"""Would a cache of hot source rows in LDS pay for NARROW rows (D = 16 / 32: the reference's default hidden width)?
A 64-byte row costs a whole 128-byte L2 request, so Reddit-like D = 16 runs at the same request rate as D = 64 with half
of every request wasted; 128 KB of LDS hold 2048 such rows (4x the 512 that fit at D = 64), i.e. the most popular 3.5 % of
a 58 K-row slice -- with power-law ids a large share of the edges.  Bare access stream (tools/ceiling/gather_ceiling.hip),
persistent 16-wavefront workgroups, ids of the `cap` most frequent rows of every slice pre-marked "read from LDS".
usage: probe_hub_narrow.py [config] [D] [phases,..] [caps,..]"""
import ctypes
import json
import os
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
from gnnadvisor_osdi21_amd import graph  # noqa: E402

lib = ctypes.CDLL(os.path.join(here, "libceiling.so"))
lib.gather_ceiling_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
lib.gather_ceiling_hub_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
dev = torch.device("cuda:0")
cfg = sys.argv[1] if len(sys.argv) > 1 else "reddit-like"
D = int(sys.argv[2]) if len(sys.argv) > 2 else 16
phases = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "2,4,8").split(",")]
caps = [int(v) for v in (sys.argv[4] if len(sys.argv) > 4 else "0,512,1024,2048").split(",")]
U = 4
g = graph.make_config_graph(cfg, device=dev)
N, nnz = g.num_nodes, g.column_index.numel()
X = torch.randn(N, D, device=dev)
col = g.column_index
slice_rows = (N + 31) // 32
out = torch.empty((nnz // 256 + 64) * 256, device=dev)
freq = torch.bincount(col.long(), minlength=N)
hub_table = torch.randn(4096, D, device=dev)


def timed(go, n=10):
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


def floor(ids, seg):
    def go():
        rc = lib.gather_ceiling_launch(X.data_ptr(), ids.data_ptr(), ids.numel(), D, seg, U, out.data_ptr())
        assert rc == 0, rc
    return timed(go)


def hub(ids, seg, hub_rows, lds_bytes):
    def go():
        rc = lib.gather_ceiling_hub_launch(X.data_ptr(), ids.data_ptr(), ids.numel(), D, seg, U, out.data_ptr(),
                                           hub_table.data_ptr(), hub_rows, lds_bytes, 256)
        assert rc == 0, rc
    return timed(go)


def marked(ids, B, cap):
    fine = torch.arange(N, device=dev) // slice_rows
    phase_of = torch.div(fine * B, 32, rounding_mode="floor")
    slot_of = torch.full((N,), -1, dtype=torch.int64, device=dev)
    for p in range(B):
        idx = (phase_of == p).nonzero().flatten()
        top = idx[torch.topk(freq[idx], min(cap, idx.numel())).indices]
        slot_of[top] = torch.arange(top.numel(), device=dev)
    s = slot_of[ids.long()]
    is_hub = s >= 0
    return torch.where(is_hub, (s | 0x80000000) - (1 << 32), ids.long()).to(torch.int32), round(float(is_hub.float().mean()), 4)


for B in phases:
    ph = torch.div((col // slice_rows) * B, 32, rounding_mode="floor").to(torch.int16)
    ids = col[torch.sort(ph, stable=True).indices].contiguous() if B > 1 else col
    del ph
    row = dict(cfg=cfg, D=D, B=B, hw_scheduled_32_waves={f"seg{seg}": floor(ids, seg) for seg in (512, 1024)}, persistent_16_waves={})
    for cap in caps:
        lds = max(cap * D * 4, 96 * 1024)                  # >= 96 KB: one workgroup per CU whatever the cache size
        if cap == 0:
            row["persistent_16_waves"]["no_cache"] = {f"seg{seg}": hub(ids, seg, 0, lds) for seg in (512, 1024)}
        else:
            mids, share = marked(ids, B, cap)
            row["persistent_16_waves"][f"cache{cap}"] = dict(from_lds=share, **{f"seg{seg}": hub(mids, seg, 0, lds) for seg in (512, 1024)})
            del mids
    print(json.dumps(row), flush=True)
    del ids
