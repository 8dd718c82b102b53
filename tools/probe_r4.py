"""Round-4 probes on the GPU box (writes gpurun_out/r4c/*.log):
  (1) torch.mm into a row-strided view (the producer-written gapped layout): correct, and no hidden copy?
  (2) wide rows in column blocks: Reddit-like D = 128 / 192 / 256, BLOCKS=2 (off) / 0 (automatic) / 1 (forced)
  (3) narrow and mid widths with and without the prepared graph: D = 16 / 32 / 64 kernel ms (regression check of the 0.4.0 refactor)
  (4) fused ReLU: relu(A X) as one call against aggregation + torch.relu
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnadvisor_osdi21_amd import _lib, graph  # noqa: E402

dev = torch.device("cuda:0")
out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r4c")
os.makedirs(out_dir, exist_ok=True)


def ms(fn, reps=20, rounds=3):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(rounds):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3 / reps)
    return best


log = open(os.path.join(out_dir, "probe_r4.log"), "w")


def say(*a):
    line = " ".join(str(x) for x in a)
    print(line, flush=True)
    log.write(line + "\n")
    log.flush()


# ---- (1) mm into a strided view
N, K, D = 232965, 602, 64
A = torch.randn(N, K, device=dev)
W = torch.randn(K, D, device=dev)
ref = torch.mm(A, W)
view = _lib.empty_rows(N, D, 128, dev)
torch.mm(A, W, out=view)
say("(1) mm into [N, 64] view with ld 128: max |diff| vs contiguous mm =", float((view - ref).abs().max()),
    "| view ptr % 512 =", view.data_ptr() % 512, "| strides", tuple(view.stride()))
cont = torch.empty(N, D, device=dev)
say("    mm contiguous out: %.4f ms | mm strided out: %.4f ms" % (ms(lambda: torch.mm(A, W, out=cont)), ms(lambda: torch.mm(A, W, out=view))))
del A, W, ref, view, cont

g = graph.make_config_graph("reddit-like", device=dev)
ps = 64
pp, p2n = _lib.build_part(ps, g.row_pointers.cpu())
ppd, p2nd = pp.to(dev), p2n.to(dev)


def kernel_ms(X, steps=10, call=None):
    out = torch.empty(X.shape[0], X.shape[1], device=dev)
    call = call or (lambda: _lib.sag(X, g.row_pointers, g.column_index, g.degrees, ppd, p2nd, ps, 32, 4, out=out))
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    _lib.profile_begin(steps)
    t0 = time.perf_counter()
    for _ in range(steps):
        call()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3 / steps
    pr = _lib.profile_end()
    return pr["main_ms"] * pr["calls"] / steps, pr["prologue_ms"] * pr["calls"] / steps, wall, _lib.last_num_phases(), _lib.last_num_launches()


# ---- (2) wide rows in column blocks
for D in (128, 160, 192, 256):
    X = torch.randn(g.num_nodes, D, device=dev)
    for blocks in (2, 0, 1):
        _lib.set_tuning(wide_blocks=blocks)
        k, p, w, ph, la = kernel_ms(X)
        say("(2) reddit-like D=%3d wide_blocks=%d: kernel %.3f ms + prologue/staging %.3f ms, wall %.3f ms, phases %d, launches %d" % (D, blocks, k, p, w, ph, la))
    _lib.reset_tuning()
    del X

# ---- (3) regression check across widths, unprepared / prepared / producer layout
for D in (16, 32, 41, 64, 100):
    X = torch.randn(g.num_nodes, D, device=dev)
    k0, p0, w0, ph0, _ = kernel_ms(X)
    _lib.prepare_graph(g.column_index, ppd, p2nd, g.num_nodes, g.num_nodes, ps, [D])
    k1, p1, w1, ph1, _ = kernel_ms(X)
    ld = _lib.preferred_ld(D, g.num_nodes, g.nnz)
    line = "(3) reddit-like D=%3d: unprepared kernel %.3f (+%.3f) wall %.3f [%d ph] | prepared kernel %.3f (+%.3f) wall %.3f [%d ph]" % (
        D, k0, p0, w0, ph0, k1, p1, w1, ph1)
    if ld != D:
        Xg = _lib.empty_rows(g.num_nodes, D, ld, dev)
        Xg.copy_(X)
        outg = torch.empty(g.num_nodes, D, device=dev)
        k2, p2, w2, ph2, _ = kernel_ms(Xg, call=lambda: _lib.agg_ld(0, Xg, g.column_index, ppd, p2nd, g.num_nodes, ps, out=outg))
        line += " | producer layout ld=%d: kernel %.3f (+%.3f) wall %.3f [%d ph]" % (ld, k2, p2, w2, ph2)
    say(line)
    _lib.release_graph(g.column_index)
    del X

# ---- (4) fused ReLU
D = 64
X = torch.randn(g.num_nodes, D, device=dev)
_lib.prepare_graph(g.column_index, ppd, p2nd, g.num_nodes, g.num_nodes, ps, [D])
out = torch.empty_like(X)
t_plain = ms(lambda: _lib.agg_ld(0, X, g.column_index, ppd, p2nd, g.num_nodes, ps, out=out))
t_fused = ms(lambda: _lib.agg_ld(0, X, g.column_index, ppd, p2nd, g.num_nodes, ps, out=out, relu=True))
t_sep = ms(lambda: torch.relu(_lib.agg_ld(0, X, g.column_index, ppd, p2nd, g.num_nodes, ps, out=out)))
say("(4) reddit-like D=64 SAG: %.3f ms | with fused ReLU: %.3f ms | aggregation + torch.relu: %.3f ms" % (t_plain, t_fused, t_sep))
_lib.set_tuning(sweep=2)
t_plain = ms(lambda: _lib.agg_ld(0, X, g.column_index, ppd, p2nd, g.num_nodes, ps, out=out))
t_fused = ms(lambda: _lib.agg_ld(0, X, g.column_index, ppd, p2nd, g.num_nodes, ps, out=out, relu=True))
say("    streaming kernel (sliced: ReLU is the whole-output pass): %.3f ms | with ReLU: %.3f ms" % (t_plain, t_fused))
_lib.reset_tuning()
