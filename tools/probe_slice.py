"""Probe: how fast is the gather when every source id falls into an X slice of a given size?
(feasibility check for the column-phased schedule: ids are folded with id % n_slice)."""
import sys, os, torch, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnadvisor_osdi21_amd import _lib, graph
dev = torch.device("cuda:0")
g = graph.make_config_graph("reddit-like", device=dev)
pp, p2n = _lib.build_part(32, g.row_pointers.cpu())
ppd, p2nd = pp.to(dev), p2n.to(dev)
D = 64
for frac in (1, 2, 4, 8, 16, 32, 64):
    n_slice = g.num_nodes // frac
    X = torch.randn(n_slice, D, device=dev)
    ci = (g.column_index % n_slice).to(torch.int32).contiguous()
    out = torch.empty(g.num_nodes, D, device=dev)
    fn = lambda: _lib.agg_rect(0, X, ci, ppd, p2nd, g.num_nodes, 32, out=out)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    _lib.profile_begin(10)
    for _ in range(10): fn()
    torch.cuda.synchronize()
    r = _lib.profile_end()
    print(json.dumps(dict(slice_MB=round(n_slice * D * 4 / 1e6, 2), ms=round(r["main_ms"], 4),
                          TBs=round(g.nnz * 260 / r["main_ms"] / 1e9, 2))), flush=True)
# host overhead probe: tiny graph, wall time per SAG call through the torch module
from gnnadvisor_osdi21_amd import load_extension
GNNA = load_extension()
gs = graph.make_config_graph("citeseer-like", device=dev)
pps, p2ns = GNNA.build_part(32, gs.row_pointers.cpu())
a = (gs.row_pointers, gs.column_index, gs.degrees, pps.to(dev), p2ns.to(dev))
Xs = torch.ones(gs.num_nodes, 16, device=dev)
for _ in range(20): GNNA.SAG(Xs, *a, 32, 32, 4)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(2000): y = GNNA.SAG(Xs, *a, 32, 32, 4)
torch.cuda.synchronize(); print("torch module SAG wall us/call (citeseer-like):", (time.perf_counter() - t0) / 2000 * 1e6)
outs = torch.empty_like(Xs)
t0 = time.perf_counter()
for _ in range(2000): _lib.sag(Xs, *a, 32, 32, 4, out=outs)
torch.cuda.synchronize(); print("ctypes C-ABI SAG wall us/call:", (time.perf_counter() - t0) / 2000 * 1e6)
t0 = time.perf_counter()
for _ in range(2000): y = torch.empty_like(Xs)
torch.cuda.synchronize(); print("empty_like us/call:", (time.perf_counter() - t0) / 2000 * 1e6)
t0 = time.perf_counter()
for _ in range(2000): y = Xs + 1
torch.cuda.synchronize(); print("torch add us/call:", (time.perf_counter() - t0) / 2000 * 1e6)
