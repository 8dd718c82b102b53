"""Probe (one GPU): the per-rank kernel work of `bench.py --gpus W` for W = 2, 4, 8 without the
collective -- rank 0's shard is generated exactly as bench.py does, the gathered feature buffer is
a random [W * n_local, D] tensor.  Prints kernel ms for the one-shot aggregation and for the
overlapped local + remote(accumulate) schedule, with the Decider-style hints and with forced
phase counts, to check the hint rules at the multi-GPU shapes."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnadvisor_osdi21_amd import _lib, graph
from gnnadvisor_osdi21_amd.dist import remap_columns_to_padded, sort_columns_within_rows, split_local_remote

dev = torch.device("cuda:0")
cfg = graph.CONFIGS["reddit-like"]
D, ps = 64, 64
n_local = cfg["num_nodes"]
e_target = int(cfg["num_edges"] * cfg.get("oversample", 1.0))
worlds = [int(a) for a in sys.argv[1:]] or [2, 4, 8]


def timed(fn, reps=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    _lib.profile_begin(reps * 2)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    r = _lib.profile_end()
    return r["main_ms"] + r["prologue_ms"]


for W in worlds:
    n_global = n_local * W
    rp, ci = graph.powerlaw_shard(n_local, n_global, e_target, min(cfg["max_degree"], n_global - 1),
                                  seed=cfg["seed"] * 1000, device=dev)
    X_all = torch.randn(n_global, D, device=dev)
    X_loc = X_all[:n_local].contiguous()
    out = torch.empty(n_local, D, device=dev)
    pp, p2n = _lib.build_part(ps, rp.cpu()); pp, p2n = pp.to(dev), p2n.to(dev)
    rp_l, ci_l, rp_r, ci_r = split_local_remote(rp, ci, 0, n_local)
    pp_l, p2n_l = _lib.build_part(ps, rp_l.cpu()); pp_l, p2n_l = pp_l.to(dev), p2n_l.to(dev)
    pp_r, p2n_r = _lib.build_part(ps, rp_r.cpu()); pp_r, p2n_r = pp_r.to(dev), p2n_r.to(dev)
    nnz, nl, nr = ci.numel(), ci_l.numel(), ci_r.numel()
    rec = {"world": W, "nnz": nnz, "nnz_local": nl, "nnz_remote": nr, "x_all_MB": round(n_global * D * 4 / 1e6, 1)}

    def whole():
        _lib.agg_rect(0, X_all, ci, pp, p2n, n_local, ps, out=out)

    def local():
        _lib.agg_rect(0, X_loc, ci_l, pp_l, p2n_l, n_local, ps, out=out)

    def remote():
        _lib.agg_rect(0, X_all, ci_r, pp_r, p2n_r, n_local, ps, out=out, accumulate=True)

    for name, fn, deg in (("whole", whole, nnz / n_local), ("local", local, nl / n_local),
                          ("remote", remote, nr / n_local)):
        _lib.reset_tuning()
        _lib.set_tuning(avg_degree=max(1, int(deg)), nonlocal_ids=1)
        ms = timed(fn)
        rec[name + "_auto"] = {"ms": round(ms, 3), "phases": _lib.last_num_phases()}
        sweep = {}
        for B in (1, 2, 4, 6, 8, 12, 16):
            _lib.set_tuning(column_phases=B)
            sweep[B] = round(timed(fn, reps=4), 3)
        rec[name + "_sweep"] = sweep
    # pipelined exchange: remote part in the sub-block-major layout, one call per source window
    for K in (2, 4, 8):
        rows_k = (n_local + K - 1) // K
        bounds = [i * n_local for i in range(W + 1)]
        ci_k = sort_columns_within_rows(rp_r, remap_columns_to_padded(ci_r, bounds, rows_k * K, K)).contiguous()
        X_k = torch.randn(W * rows_k * K, D, device=dev)
        _lib.reset_tuning()
        _lib.set_tuning(avg_degree=max(1, int(nr / n_local)), nonlocal_ids=1)

        def windows():
            for k in range(K):
                _lib.agg_rect(0, X_k, ci_k, pp_r, p2n_r, n_local, ps, out=out, accumulate=True, windows=(K, k, k + 1))
        for _ in range(2):
            windows()
        torch.cuda.synchronize()
        _lib.profile_begin(8 * K)
        for _ in range(8):
            windows()
        torch.cuda.synchronize()
        r = _lib.profile_end()
        rec[f"remote_windows_K{K}"] = {"ms": round((r["main_ms"] + r["prologue_ms"]) * K, 3), "phases": _lib.last_num_phases()}
        # round 2: the remote part split into K small CSRs (one per exchange piece), K ordinary accumulate calls
        win = (W * rows_k * K) // K
        rows_r = torch.repeat_interleave(torch.arange(n_local, device=dev), (rp_r[1:] - rp_r[:-1]).long())
        piece_of = torch.div(ci_k.long(), win, rounding_mode="floor")
        pieces = []
        for k in range(K):
            m = piece_of == k
            rp_p = torch.zeros(n_local + 1, dtype=torch.int64, device=dev)
            rp_p[1:] = torch.cumsum(torch.bincount(rows_r[m], minlength=n_local), 0)
            pp_p, p2n_p = _lib.build_part(ps, rp_p.to(torch.int32).cpu())
            pieces.append(((ci_k[m] - k * win).to(torch.int32).contiguous(), pp_p.to(dev), p2n_p.to(dev)))
            _lib.set_graph_hints(pieces[-1][0], max(1, int(nr / n_local / K)), True)

        def piecewise():
            for k, (ci_p, pp_p, p2n_p) in enumerate(pieces):     # ids relative to the piece's window of the buffer
                _lib.agg_rect(0, X_k[k * win:(k + 1) * win], ci_p, pp_p, p2n_p, n_local, ps, out=out, accumulate=True)
        rec[f"remote_pieces_K{K}"] = {"ms": round(timed(piecewise) * K, 3), "phases_last": _lib.last_num_phases()}
        sweep = {}
        for B in (1, 2, 4, 8, 16):
            _lib.set_tuning(column_phases=B)
            sweep[B] = round(timed(piecewise, reps=4) * K, 3)
        rec[f"remote_pieces_K{K}_sweep"] = sweep
        _lib.reset_tuning()
        del X_k, ci_k, pieces
    _lib.reset_tuning()
    rec["edges_per_s_overlap_kernels_only"] = nnz / ((rec["local_auto"]["ms"] + rec["remote_auto"]["ms"]) * 1e-3)
    print(json.dumps(rec), flush=True)
    del X_all, X_loc, out
    torch.cuda.empty_cache()
