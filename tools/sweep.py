#!/usr/bin/env python3
"""Scheduler-knob sweep for the aggregation kernel on one GPU (used to choose the defaults
documented in DESIGN.md "Tuning").  Prints one line per configuration:
kernel ms (HIP events, gnna_profile_*), G edges/s, gather-model TB/s."""
import argparse
import itertools
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnadvisor_osdi21_amd import _lib, graph  # noqa: E402


def time_cfg(fn, steps=10, warmup=3, events=False):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if events:   # kernels without library-side event hooks (sddmm): torch events on the current stream
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(steps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return dict(main_ms=a.elapsed_time(b) / steps, prologue_ms=0.0, calls=steps)
    _lib.profile_begin(steps)
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return _lib.profile_end()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="reddit-like")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--locality", type=float, default=0.0)
    ap.add_argument("--dims", default="64")
    ap.add_argument("--ps", default="32")
    ap.add_argument("--G", default="16")
    ap.add_argument("--U", default="8")
    ap.add_argument("--bpc", default="0")
    ap.add_argument("--xcd", default="1")
    ap.add_argument("--trust", default="0")
    ap.add_argument("--phases", default="0")
    ap.add_argument("--prescale", default="0")
    ap.add_argument("--mode", default="sag")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--sources-mult", type=int, default=1,
                    help="destination shard of a graph with this many times more source nodes (rect kernel)")
    args = ap.parse_args()
    ints = lambda s: [int(v) for v in s.split(",")]
    dev = torch.device("cuda:0")
    if args.sources_mult > 1:
        c = graph.CONFIGS[args.config]
        n = int(c["num_nodes"] * args.scale)
        rp, ci = graph.powerlaw_shard(n, n * args.sources_mult, int(c["num_edges"] * args.scale * c.get("oversample", 1.0)),
                                      c["max_degree"], seed=c["seed"], device=dev)
        g = graph.CSRGraph(n, rp, ci, graph.degrees_from_rowptr(rp), int(ci.numel()), ci.numel() / n, 0.0)
    else:
        g = graph.make_config_graph(args.config, device=dev, locality=args.locality, scale=args.scale)
    n_src = g.num_nodes * args.sources_mult
    rp_cpu = g.row_pointers.cpu()
    print(f"# graph {args.config}: N={g.num_nodes} nnz={g.nnz}", flush=True)
    for ps in ints(args.ps):
        pp, p2n = _lib.build_part(ps, rp_cpu)
        ppd, p2nd = pp.to(dev), p2n.to(dev)
        P = p2n.numel()
        for D in ints(args.dims):
            X = torch.randn(n_src, D, device=dev)
            out = torch.empty(g.num_nodes, D, device=dev)
            bytes_ = g.nnz * (4 * D + 4) + g.num_nodes * (4 * D + 4) + P * 8
            for G, U, bpc, xcd, trust, ph, pre in itertools.product(ints(args.G), ints(args.U), ints(args.bpc),
                                                                    ints(args.xcd), ints(args.trust), ints(args.phases),
                                                                    ints(args.prescale)):
                _lib.set_tuning(G, U, bpc, xcd, trust, ph, gcn_prescale=pre)
                if args.sources_mult > 1:
                    fn = lambda: _lib.agg_rect(0, X, g.column_index, ppd, p2nd, g.num_nodes, ps, out=out)
                elif args.mode == "sag":
                    fn = lambda: _lib.sag(X, g.row_pointers, g.column_index, g.degrees, ppd, p2nd, ps, 32, 4, out=out)
                elif args.mode == "sddmm":
                    eo = torch.empty(g.nnz, device=dev)
                    fn = lambda: _lib.sddmm(X, X, g.column_index, ppd, p2nd, ps, out=eo)
                elif args.mode == "gcn":
                    fn = lambda: _lib.agg_gcn(X, g.row_pointers, g.column_index, g.degrees, ppd, p2nd, ps, 32, 4, out=out)
                else:
                    fn = lambda: _lib.agg_gin(X, g.row_pointers, g.column_index, 0.5, ppd, p2nd, ps, 32, 4, out=out)
                r = time_cfg(fn, args.steps, events=(args.mode == "sddmm"))
                ms = r["main_ms"]
                print(json.dumps(dict(ps=ps, D=D, G=G, U=U, bpc=bpc, xcd=xcd, trust=trust, ph=ph, pre=pre, P=P,
                                      ms=round(ms, 4), pro_ms=round(r["prologue_ms"], 4),
                                      Gedges=round(g.nnz / ms / 1e6, 2),
                                      TBs=round(bytes_ / ms / 1e9, 3))), flush=True)
    _lib.reset_tuning()


if __name__ == "__main__":
    main()
