"""Does the gather care where a 256-byte row starts?  The bare access stream (tools/ceiling) over the Reddit-like
headline's ids, phase-major, with the rows (a) packed back to back as in X, (b) every row at an EVEN 256-byte position
(ids doubled: a row starts on a 512-byte boundary, the odd positions are never touched), (c) every row at an ODD position.
usage: probe_row_alignment.py   (build first: tools/ceiling/build.sh)"""
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
g = graph.make_config_graph("reddit-like", device=dev)
N, nnz, D = g.num_nodes, g.column_index.numel(), 64
X2 = torch.randn(2 * N + 2, D, device=dev)
col = g.column_index
out = torch.empty((nnz // 256 + 64) * 256, device=dev)


def floor_ms(ids, seg=512, U=4, n=10):
    def go():
        assert lib.gather_ceiling_launch(X2.data_ptr(), ids.data_ptr(), ids.numel(), D, seg, U, out.data_ptr()) == 0
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


for B in (1, 8, 16, 32):
    ph = torch.div(col.long() * B, N, rounding_mode="floor").to(torch.int16)
    ids = col[torch.sort(ph, stable=True).indices].contiguous() if B > 1 else col
    print(json.dumps(dict(B=B, back_to_back=floor_ms(ids), even_positions=floor_ms((ids * 2).contiguous()),
                          odd_positions=floor_ms((ids * 2 + 1).contiguous()))), flush=True)
