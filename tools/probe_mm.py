"""Probe: what torch.mm (hipBLASLt / rocBLAS fp32) achieves on the dense-update shapes of the GCN /
GIN layers (tall-skinny: M = nodes, K, N <= a few hundred) against the memory-bound ideal."""
import sys, os, json, torch
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
                (410236, 96, 16), (410236, 16, 22)):
    X = torch.randn(M, K, device=dev); W = torch.randn(K, N, device=dev); G = torch.randn(M, N, device=dev)
    fwd = t(lambda: torch.mm(X, W))                 # update
    dx = t(lambda: torch.mm(G, W.t()))              # d_input
    dw = t(lambda: torch.mm(X.t(), G))              # d_weight
    ideal_fwd = (M * K + M * N) * 4 / 6.0e12 * 1e3
    print(json.dumps(dict(M=M, K=K, N=N, fwd_ms=round(fwd, 3), dx_ms=round(dx, 3), dw_ms=round(dw, 3),
                          ideal_fwd_ms=round(ideal_fwd, 3), fwd_TFLOPs=round(2 * M * K * N / fwd / 1e9, 1))), flush=True)
