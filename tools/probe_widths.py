"""Kernel ms over row widths on one graph (prepared, producer-written layout): `python tools/probe_widths.py [config] [ps]`.
The question: do the odd widths of a last layer (D = number of classes: 41, 47, 7 ...) cost what their bytes say?"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnadvisor_osdi21_amd import _lib, graph  # noqa: E402

dev = torch.device("cuda:0")
cfg = sys.argv[1] if len(sys.argv) > 1 else "reddit-like"
ps = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dims = [int(d) for d in sys.argv[3].split(",")] if len(sys.argv) > 3 else [16, 24, 32, 40, 41, 44, 47, 48, 56, 64]
g = graph.make_config_graph(cfg, device=dev)
pp, p2n = _lib.build_part(ps, g.row_pointers.cpu())
ppd, p2nd = pp.to(dev), p2n.to(dev)
N, E = g.num_nodes, int(g.column_index.numel())
_lib.prepare_graph(g.column_index, ppd, p2nd, N, N, ps, dims)
for D in dims:
    ld = _lib.preferred_ld(D, N, E)
    X = _lib.empty_rows(N, D, ld, dev)
    X.copy_(torch.randn(N, D, device=dev))
    out = torch.empty(N, D, device=dev)
    steps = 20
    for _ in range(3):
        _lib.agg_ld(0, X, g.column_index, ppd, p2nd, N, ps, out=out)
    torch.cuda.synchronize()
    _lib.profile_begin(steps)
    for _ in range(steps):
        _lib.agg_ld(0, X, g.column_index, ppd, p2nd, N, ps, out=out)
    torch.cuda.synchronize()
    pr = _lib.profile_end()
    print(json.dumps(dict(cfg=cfg, D=D, ld=ld, ps=ps, kernel_ms=round(pr["main_ms"], 4), prologue_ms=round(pr.get("prologue_ms", 0.0), 4),
                          phases=_lib.last_num_phases(), G_edges_s=round(E / pr["main_ms"] / 1e6, 1),
                          ms_per_64_floats=round(pr["main_ms"] * 64 / D, 4))), flush=True)
