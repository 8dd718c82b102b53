"""Probe: libgnna's X^T G kernel against torch.mm(X.t(), G) on the layer shapes."""
import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnadvisor_osdi21_amd import _lib
dev = torch.device("cuda:0")


def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for M, K, N in ((2449029, 64, 64), (2449029, 100, 64), (2449029, 64, 47), (232965, 602, 64), (232965, 64, 41),
                (410236, 96, 16), (410236, 16, 22), (2708, 1433, 16), (13882495, 128, 128)):
    X = torch.randn(M, K, device=dev); G = torch.randn(M, N, device=dev)
    out = torch.empty(K, N, device=dev)
    ours = t(lambda: _lib.xtg(X, G, out=out)); blas = t(lambda: torch.mm(X.t(), G))
    err = float((out - torch.mm(X.t(), G)).abs().max() / torch.mm(X.t().abs(), G.abs()).max())
    print(json.dumps(dict(M=M, K=K, N=N, xtg_ms=round(ours, 3), torch_mm_ms=round(blas, 3),
                          ideal_ms=round((M * K + M * N) * 4 / 6.0e12 * 1e3, 3), rel_err=err)), flush=True)
