"""Probe: does renumbering the nodes by descending degree (hot source rows packed into a small
address range: fewer pages / better cache-set use) speed up the gather on a graph without
community structure?  products-like and reddit-like, D = 64."""
import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnadvisor_osdi21_amd import _lib, graph
dev = torch.device("cuda:0")


def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    _lib.profile_begin(reps)
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    r = _lib.profile_end()
    return r["main_ms"]


for name, ps in (("products-like", 32), ("reddit-like", 64)):
    g = graph.make_config_graph(name, device=dev)
    n = g.num_nodes
    rows = torch.repeat_interleave(torch.arange(n, device=dev), (g.row_pointers[1:] - g.row_pointers[:-1]).long())
    cols = g.column_index.long()
    deg = (g.row_pointers[1:] - g.row_pointers[:-1]).long()
    variants = {"as generated": None,
                "degree-descending": torch.argsort(deg, descending=True),
                "degree-ascending": torch.argsort(deg)}
    X = torch.randn(n, 64, device=dev)
    for vname, order in variants.items():
        if order is None:
            gg = g
        else:
            new_id = torch.empty_like(order); new_id[order] = torch.arange(n, device=dev)
            gg = graph.graph_from_edges(new_id[rows], new_id[cols], n)
        pp, p2n = _lib.build_part(ps, gg.row_pointers.cpu()); pp, p2n = pp.to(dev), p2n.to(dev)
        out = torch.empty(n, 64, device=dev)
        res = {}
        for hints in (False, True):
            _lib.reset_tuning(); _lib.set_graph_hints(None, 0, False)
            if hints:
                _lib.set_graph_hints(gg.column_index, gg.nnz / n, True)
            res["hints" if hints else "plain"] = (round(timed(lambda: _lib.sag(X, gg.row_pointers, gg.column_index, gg.degrees, pp, p2n, ps, 32, 4, out=out)), 3), _lib.last_num_phases())
        print(json.dumps(dict(graph=name, order=vname, ms=res)), flush=True)
