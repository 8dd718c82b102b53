"""Floor of the sliced gather when the slices are POPULARITY CLASSES of source rows instead of id ranges: the rows are
ranked by how often they are gathered (column frequency); the hottest `hot_rows` rows form slice 0, the next `hot_rows`
slice 1, ... for `n_hot` slices (each small enough to live in an XCD's 4 MiB L2), the remaining rows are split into
`n_cold` equal slices.  The bare access stream (tools/ceiling/gather_ceiling.hip) walks the ids phase-major as the
kernel would.  Rows stay where they are in memory -- only the ORDER in which edges are consumed changes.
usage: probe_popularity_slices.py [config] [D]   (build first: tools/ceiling/build.sh)"""
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
dev = torch.device("cuda:0")
cfg = sys.argv[1] if len(sys.argv) > 1 else "reddit-like"
D = int(sys.argv[2]) if len(sys.argv) > 2 else 64
g = graph.make_config_graph(cfg, device=dev)
N, nnz = g.num_nodes, g.column_index.numel()
X = torch.randn(N, D, device=dev)
col = g.column_index
out = torch.empty((nnz // 256 + 64) * 256, device=dev)
freq = torch.bincount(col.long(), minlength=N)
rank = torch.empty(N, dtype=torch.int64, device=dev)
rank[torch.argsort(freq, descending=True)] = torch.arange(N, device=dev)
row_bytes = D * 4


def floor_ms(ids, seg=512, U=4, n=10):
    def go():
        assert lib.gather_ceiling_launch(X.data_ptr(), ids.data_ptr(), ids.numel(), D, seg, U, out.data_ptr()) == 0
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


def run(name, phase_of_row, B):
    ph = phase_of_row[col.long()].to(torch.int16)
    ids = col[torch.sort(ph, stable=True).indices].contiguous()
    share = torch.bincount(ph.long(), minlength=B).double() / nnz
    print(json.dumps(dict(cfg=cfg, D=D, slices=name, B=B, floor_ms=floor_ms(ids),
                          edge_share_per_slice=[round(float(v), 3) for v in share])), flush=True)


# today's slices: equal id ranges
for B in (8, 16):
    run(f"{B} id ranges", torch.div(torch.arange(N, device=dev) * B, N, rounding_mode="floor"), B)
# popularity classes
for hot_mb, n_hot, n_cold in ((3.5, 5, 3), (3.5, 3, 5), (3.0, 6, 2), (3.5, 4, 4), (2.0, 8, 8), (3.5, 8, 8), (3.5, 2, 2), (3.5, 3, 3),
                              (3.5, 4, 2), (7.4, 4, 4)):
    hot_rows = int(hot_mb * 1e6 / row_bytes)
    B = n_hot + n_cold
    cold_rows = max(1, (N - hot_rows * n_hot + n_cold - 1) // n_cold)
    phase = torch.where(rank < hot_rows * n_hot, torch.div(rank, hot_rows, rounding_mode="floor"),
                        n_hot + torch.div((rank - hot_rows * n_hot).clamp(min=0), cold_rows, rounding_mode="floor")).clamp(max=B - 1)
    run(f"{n_hot} hot x {hot_mb} MB + {n_cold} cold", phase, B)
