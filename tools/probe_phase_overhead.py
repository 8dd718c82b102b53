"""Probe: what do the column phases cost when the gather is L2-resident anyway?  Ids of the
Reddit-like graph are folded into a 3.7 MB slice of X (id % n_slice, re-sorted inside rows);
time vs number of phases isolates the per-phase overhead from the hit-rate gain."""
import sys, os, torch, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnadvisor_osdi21_amd import _lib, graph
from gnnadvisor_osdi21_amd.dist import sort_columns_within_rows
dev = torch.device("cuda:0")
g = graph.make_config_graph("reddit-like", device=dev)
D = 64
for ps in (64, 32):
    pp, p2n = _lib.build_part(ps, g.row_pointers.cpu())
    ppd, p2nd = pp.to(dev), p2n.to(dev)
    for frac in (16, 1):
        n_slice = g.num_nodes // frac
        X = torch.randn(n_slice, D, device=dev)
        ci = sort_columns_within_rows(g.row_pointers, (g.column_index % n_slice).to(torch.int32)).contiguous()
        out = torch.empty(g.num_nodes, D, device=dev)
        res = {}
        for B in (1, 2, 4, 8, 16):
            _lib.set_tuning(column_phases=B)
            fn = lambda: _lib.agg_rect(0, X, ci, ppd, p2nd, g.num_nodes, ps, out=out)
            for _ in range(3): fn()
            torch.cuda.synchronize()
            _lib.profile_begin(10)
            for _ in range(10): fn()
            torch.cuda.synchronize()
            res[B] = round(_lib.profile_end()["main_ms"], 4)
        _lib.reset_tuning()
        print(json.dumps(dict(partSize=ps, slice_MB=round(n_slice * D * 4 / 1e6, 2), ms_by_phases=res)), flush=True)
