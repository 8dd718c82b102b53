"""Floor of the sliced gather on this chip: the bare access stream (tools/ceiling/gather_ceiling.hip) of the
Reddit-like headline in the streaming kernel's order, for B phases, against the kernels themselves.
usage: probe_ceiling.py [config] [D] [phases,..] [seg,..] [U,..]   (build first: tools/ceiling/build.sh)"""
import ctypes
import json
import os
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
from gnnadvisor_osdi21_amd import _lib, graph  # noqa: E402

lib = ctypes.CDLL(os.path.join(here, "libceiling.so"))
lib.gather_ceiling_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
cfg = sys.argv[1] if len(sys.argv) > 1 else "reddit-like"
D = int(sys.argv[2]) if len(sys.argv) > 2 else 64
phases = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "1,4,8,16,32").split(",")]
segs = [int(v) for v in (sys.argv[4] if len(sys.argv) > 4 else "256,512,4096").split(",")]
Us = [int(v) for v in (sys.argv[5] if len(sys.argv) > 5 else "4,8").split(",")]
g = graph.make_config_graph(cfg, device=dev)
N, nnz = g.num_nodes, g.column_index.numel()
X = torch.randn(N, D, device=dev)
col = g.column_index
slice_rows = (N + 31) // 32
out = torch.empty((nnz // 256 + 64) * 256, device=dev)


def time_launch(ids, seg, U, n=10):
    def go():
        rc = lib.gather_ceiling_launch(X.data_ptr(), ids.data_ptr(), ids.numel(), D, seg, U, out.data_ptr())
        assert rc == 0, rc
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


lib.gather_ceiling_hub_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
freq = torch.bincount(col.long(), minlength=N)
hub_table = torch.randn(1024, D, device=dev)
hub_caps = [int(v) for v in os.environ.get("HUB_CAPS", "0,256,384,512").split(",")]


def time_hub(ids, seg, U, hub_rows, lds_bytes, blocks=256, n=10):
    def go():
        rc = lib.gather_ceiling_hub_launch(X.data_ptr(), ids.data_ptr(), ids.numel(), D, seg, U, out.data_ptr(),
                                           hub_table.data_ptr(), hub_rows, lds_bytes, blocks)
        assert rc == 0, rc
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


def marked(ids, B, cap):
    """the `cap` most frequent ids of every phase's id range replaced by 0x80000000 | slot"""
    fine = torch.arange(N, device=dev) // slice_rows
    phase_of = torch.div(fine * B, 32, rounding_mode="floor")
    slot_of = torch.full((N,), -1, dtype=torch.int64, device=dev)
    for p in range(B):
        idx = (phase_of == p).nonzero().flatten()
        top = idx[torch.topk(freq[idx], min(cap, idx.numel())).indices]
        slot_of[top] = torch.arange(top.numel(), device=dev)
    s = slot_of[ids.long()]
    hub = s >= 0
    return torch.where(hub, (s | 0x80000000) - (1 << 32), ids.long()).to(torch.int32), round(float(hub.float().mean()), 4)


for B in phases:
    if B == 1:
        ids = col
    else:
        ph = torch.div((col // slice_rows) * B, 32, rounding_mode="floor").to(torch.int16)
        ids = col[torch.sort(ph, stable=True).indices].contiguous()
        del ph
    row = dict(cfg=cfg, D=D, B=B, floor_ms={})
    for seg in segs:
        for U in Us:
            row["floor_ms"][f"seg{seg}_U{U}"] = time_launch(ids, seg, U)
    row["persistent_16_waves_per_cu"] = {}
    for cap in (hub_caps if D == 64 else []):
        lds = max(cap * D * 4, 96 * 1024)                  # >= 96 KB: one workgroup per CU whatever the cache size
        if cap == 0:
            row["persistent_16_waves_per_cu"]["no_cache"] = {f"seg{seg}": time_hub(ids, seg, 4, 0, lds) for seg in segs}
        else:
            mids, share = marked(ids, B, cap)
            row["persistent_16_waves_per_cu"][f"cache{cap}"] = dict(from_lds=share, **{f"seg{seg}": time_hub(mids, seg, 4, 0, lds) for seg in segs})
            del mids
    print(json.dumps(row), flush=True)
    del ids
