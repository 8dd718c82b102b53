"""Probe: keep the destination (row) order, but store the source feature rows in degree-descending
order (column ids remapped): hot source rows become one compact region.  products-like, D = 64."""
import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnadvisor_osdi21_amd import _lib, graph
from gnnadvisor_osdi21_amd.dist import sort_columns_within_rows
dev = torch.device("cuda:0")


def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    _lib.profile_begin(reps)
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return _lib.profile_end()["main_ms"]


for name, ps in (("products-like", 32), ("reddit-like", 64)):
    g = graph.make_config_graph(name, device=dev)
    n = g.num_nodes
    deg = (g.row_pointers[1:] - g.row_pointers[:-1]).long()
    pp, p2n = _lib.build_part(ps, g.row_pointers.cpu()); pp, p2n = pp.to(dev), p2n.to(dev)
    X = torch.randn(n, 64, device=dev)
    out = torch.empty(n, 64, device=dev)
    for vname in ("as generated", "sources degree-descending", "sources random"):
        if vname == "as generated":
            ci = g.column_index
        else:
            order = torch.argsort(deg, descending=True) if "degree" in vname else torch.randperm(n, device=dev)
            new_id = torch.empty_like(order); new_id[order] = torch.arange(n, device=dev)
            ci = sort_columns_within_rows(g.row_pointers, new_id[g.column_index.long()].to(torch.int32)).contiguous()
        res = {}
        for B in (1, 2, 4, 8):
            _lib.reset_tuning(); _lib.set_tuning(column_phases=B)
            res[B] = round(timed(lambda: _lib.agg_rect(0, X, ci, pp, p2n, n, ps, out=out)), 3)
        _lib.reset_tuning()
        print(json.dumps(dict(graph=name, layout=vname, ms_by_phases=res)), flush=True)
