"""Probe: PyTorch TunableOp (times the rocBLAS / hipBLASLt solutions per GEMM shape at first use) on the
dense-update shapes."""
import json, time, torch
dev = torch.device("cuda:0")


def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


shapes = ((2449029, 64, 64), (2449029, 100, 64), (232965, 602, 64), (232965, 64, 41), (410236, 96, 16))
base = {}
for M, K, N in shapes:
    X = torch.randn(M, K, device=dev); W = torch.randn(K, N, device=dev); G = torch.randn(M, N, device=dev)
    base[(M, K, N)] = (t(lambda: torch.mm(X, W)), t(lambda: torch.mm(G, W.t())))
import torch.cuda.tunable as tun
tun.enable(True); tun.tuning_enable(True)
try:
    tun.write_file_on_exit(False)
except Exception as e:
    print("write_file_on_exit:", e)
try:
    tun.set_max_tuning_duration(200); tun.set_max_tuning_iterations(20)
except Exception as e:
    print("limits:", e)
for M, K, N in shapes:
    X = torch.randn(M, K, device=dev); W = torch.randn(K, N, device=dev); G = torch.randn(M, N, device=dev)
    t0 = time.perf_counter(); torch.mm(X, W); torch.mm(G, W.t()); torch.cuda.synchronize(); tune_s = time.perf_counter() - t0
    f, d = t(lambda: torch.mm(X, W)), t(lambda: torch.mm(G, W.t()))
    print(json.dumps(dict(M=M, K=K, N=N, fwd_ms=[round(base[(M, K, N)][0], 3), round(f, 3)],
                          dx_ms=[round(base[(M, K, N)][1], 3), round(d, 3)], tuning_s=round(tune_s, 2),
                          ideal_ms=round((M * K + M * N) * 4 / 6.0e12 * 1e3, 3))), flush=True)
