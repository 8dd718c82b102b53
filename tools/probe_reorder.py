"""Probe: aggregation time on a Reddit-like graph with hidden locality (90 % of the edges within +-4096 ids)
whose ids were scrambled, before and after the native community renumbering; hidden order for reference."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnadvisor_osdi21_amd import _lib, graph

dev = torch.device("cuda:0")
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
D = int(sys.argv[2]) if len(sys.argv) > 2 else 64
wrap = (sys.argv[3] if len(sys.argv) > 3 else "ring") == "ring"
g = graph.make_config_graph("reddit-like", device=dev, locality=0.9, scale=scale, wrap=wrap)
print(json.dumps({"hidden_topology": "ring" if wrap else "line"}))
n = g.num_nodes
rows = torch.repeat_interleave(torch.arange(n, device=dev), (g.row_pointers[1:] - g.row_pointers[:-1]).long())
cols = g.column_index.long()


def timed(rp, ci, tag, ps=64):
    pp, p2n = _lib.build_part(ps, rp.cpu())
    ppd, p2nd = pp.to(dev), p2n.to(dev)
    X = torch.randn(n, D, device=dev)
    out = torch.empty_like(X)
    res = {}
    for name, kw in (("auto", {}), ("single_pass", dict(column_phases=1))):
        _lib.reset_tuning(); _lib.set_tuning(**kw)
        fn = lambda: _lib.sag(X, rp, ci, None, ppd, p2nd, ps, 32, 4, out=out)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        _lib.profile_begin(10)
        for _ in range(10): fn()
        torch.cuda.synchronize()
        res[name] = (round(_lib.profile_end()["main_ms"], 4), _lib.last_num_phases())
    _lib.reset_tuning()
    print(json.dumps({"order": tag, "ms(phases)": res}), flush=True)


timed(g.row_pointers, g.column_index, "hidden (generator's)")
perm = torch.randperm(n, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
src, dst = perm[rows], perm[cols]
rp_s, ci_s = graph.csr_from_edges(src, dst, n)
timed(rp_s, ci_s, "scrambled")
src_c, dst_c = src.cpu(), dst.cpu()
for name, fn in (("community", _lib.reorder_community), ("rcm", _lib.reorder_rcm)):
    t0 = time.time()
    new_id = fn(src_c, dst_c, n).to(dev).long()
    dt = time.time() - t0
    rp_r, ci_r = graph.csr_from_edges(new_id[src], new_id[dst], n)
    print(json.dumps({"reorder": name, "seconds": round(dt, 1), "span_before": _lib.edge_span(src_c, dst_c),
                      "span_after": _lib.edge_span(new_id[src].cpu(), new_id[dst].cpu())}), flush=True)
    timed(rp_r, ci_r, name)
