"""Probe: torch.mm on the dense-update shapes with hipBLASLt vs rocBLAS as the preferred library."""
import json, torch
dev = torch.device("cuda:0")


def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for lib in ("hipblaslt", "rocblas" if hasattr(torch.backends.cuda, "preferred_blas_library") else None):
    if lib is None:
        continue
    try:
        torch.backends.cuda.preferred_blas_library(lib if lib != "rocblas" else "cublas")
    except Exception as e:
        print("cannot select", lib, e); continue
    for M, K, N in ((2449029, 64, 64), (2449029, 100, 64), (232965, 602, 64), (232965, 64, 41), (410236, 96, 16)):
        X = torch.randn(M, K, device=dev); W = torch.randn(K, N, device=dev); G = torch.randn(M, N, device=dev)
        print(json.dumps(dict(lib=lib, M=M, K=K, N=N, fwd_ms=round(t(lambda: torch.mm(X, W)), 3),
                              dx_ms=round(t(lambda: torch.mm(G, W.t())), 3),
                              ideal_ms=round((M * K + M * N) * 4 / 6.0e12 * 1e3, 3))), flush=True)
