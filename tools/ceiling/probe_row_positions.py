"""Where should a 256-byte row start?  Follow-up of probe_row_alignment.py: the bare access stream of the Reddit-like
headline (16 source slices, ids phase-major) with row `id` placed at byte  (id * S + off) * (4 D) + shift  for several
(S, off, shift).  usage: probe_row_positions.py [slices] [D = 64 | 128 | 32]   (build first: tools/ceiling/build.sh)"""
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
D = int(sys.argv[2]) if len(sys.argv) > 2 else 64
N, nnz = g.num_nodes, g.column_index.numel()
XB = torch.randn((16 if D <= 64 else 4) * N + 64, D, device=dev)
col = g.column_index
out = torch.empty((nnz // 256 + 64) * 256, device=dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ph = torch.div(col.long() * B, N, rounding_mode="floor").to(torch.int16)
ids0 = col[torch.sort(ph, stable=True).indices].contiguous()
print("base % 4096 =", XB.data_ptr() % 4096, flush=True)


def floor_ms(ids, shift=0, seg=512, U=4, n=10):
    def go():
        assert lib.gather_ceiling_launch(XB.data_ptr() + shift, ids.data_ptr(), ids.numel(), D, seg, U, out.data_ptr()) == 0
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


cases = [(1, 0, 0), (2, 0, 0), (2, 1, 0), (2, 0, 0), (2, 1, 0), (4, 0, 0), (4, 1, 0), (4, 2, 0), (4, 3, 0), (8, 0, 0), (8, 4, 0), (8, 2, 0), (16, 0, 0),
         (3, 0, 0), (2, 0, 128), (2, 1, 128), (1, 0, 128), (1, 0, 64), (2, 0, 64)]
if D > 64:
    cases = [(1, 0, 0), (2, 0, 0), (2, 1, 0), (4, 0, 0), (4, 1, 0), (4, 2, 0), (4, 3, 0), (1, 0, 128), (1, 0, 256), (2, 0, 256), (2, 1, 256), (1, 0, 0)]
elif D < 64:
    cases = [(1, 0, 0), (2, 0, 0), (2, 1, 0), (4, 0, 0), (4, 1, 0), (4, 2, 0), (4, 3, 0), (8, 0, 0), (8, 2, 0), (8, 4, 0), (8, 6, 0), (8, 1, 0), (1, 0, 64)]
if D == 16:
    # 64-byte rows, two per line: contiguous against layouts that leave the slow line class (the fourth line of every 512 bytes) empty
    U16 = 2 if len(sys.argv) > 3 and sys.argv[3] == "u2" else 4
    for name, ids in (("contiguous", ids0), ("6 of every 8 row slots (bytes 384..511 of every 512 empty)", (ids0 // 6) * 8 + ids0 % 6),
                      ("14 of every 16 row slots (bytes 384..511 of every 1024 empty)", (ids0 // 14) * 16 + (ids0 % 14) + 2 * ((ids0 % 14) >= 6).int()),
                      ("one row per 128-byte line", ids0 * 2), ("contiguous again", ids0)):
        print(json.dumps(dict(D=D, B=B, layout=name, ms=floor_ms(ids.contiguous().int(), 0, U=U16))), flush=True)
    sys.exit(0)
for S, off, shift in cases:
    ids = (ids0 * S + off).contiguous()
    print(json.dumps(dict(D=D, B=B, S=S, off=off, shift=shift, ms=floor_ms(ids, shift))), flush=True)
