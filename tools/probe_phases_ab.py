"""Probe: kernel time vs number of column phases on the Reddit-like graph (D = 64), (a) as is and
(b) with the ids folded into a 3.7 MB slice of X (L2 resident: isolates the per-phase overhead
from the hit-rate gain).  Run once per library build (GNNA_LIB=... selects an A/B build)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnadvisor_osdi21_amd import _lib, graph  # noqa: E402
from gnnadvisor_osdi21_amd.dist import sort_columns_within_rows  # noqa: E402

dev = torch.device("cuda:0")
cfg = sys.argv[1] if len(sys.argv) > 1 else "reddit-like"
D = int(sys.argv[2]) if len(sys.argv) > 2 else 64
phases = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "1,2,4,6,8,12,16").split(",")]
Gs = [int(v) for v in (sys.argv[4] if len(sys.argv) > 4 else "16,32").split(",")]
fold = int(sys.argv[5]) if len(sys.argv) > 5 else 16
g = graph.make_config_graph(cfg, device=dev)
ps = 64
pp, p2n = _lib.build_part(ps, g.row_pointers.cpu())
ppd, p2nd = pp.to(dev), p2n.to(dev)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    _lib.profile_begin(n)
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return round(_lib.profile_end()["main_ms"], 4)


out = torch.empty(g.num_nodes, D, device=dev)
X = torch.randn(g.num_nodes, D, device=dev)
for G in Gs:
    res = {}
    for B in phases:
        _lib.set_tuning(groups_per_chunk=G, column_phases=B)
        res[B] = timeit(lambda: _lib.sag(X, g.row_pointers, g.column_index, g.degrees, ppd, p2nd, ps, 32, 4, out=out))
    print(json.dumps(dict(lib=os.path.basename(_lib.LIB_PATH), cfg=cfg, D=D, G=G, folded=False, ms_by_phases=res)), flush=True)
if fold > 1:
    n_slice = g.num_nodes // fold
    Xs = torch.randn(n_slice, D, device=dev)
    ci = sort_columns_within_rows(g.row_pointers, (g.column_index % n_slice).to(torch.int32)).contiguous()
    for G in Gs:
        res = {}
        for B in phases:
            _lib.set_tuning(groups_per_chunk=G, column_phases=B)
            res[B] = timeit(lambda: _lib.agg_rect(0, Xs, ci, ppd, p2nd, g.num_nodes, ps, out=out))
        print(json.dumps(dict(lib=os.path.basename(_lib.LIB_PATH), cfg=cfg, D=D, G=G, folded=True,
                              slice_MB=round(n_slice * D * 4 / 1e6, 2), ms_by_phases=res)), flush=True)
    # ids folded into the TOP slice: all phases but the last find nothing to consume -> the fixed
    # cost of a phase (launch, descriptors, cursors, one id tile per run) without any gather work
    ci_top = (ci + (g.num_nodes - n_slice)).contiguous()
    for G in Gs:
        res = {}
        for B in phases:
            _lib.set_tuning(groups_per_chunk=G, column_phases=B)
            res[B] = timeit(lambda: _lib.sag(X, g.row_pointers, ci_top, g.degrees, ppd, p2nd, ps, 32, 4, out=out))
        print(json.dumps(dict(lib=os.path.basename(_lib.LIB_PATH), cfg=cfg, D=D, G=G, folded="top (empty phases)",
                              ms_by_phases=res)), flush=True)
_lib.reset_tuning()
