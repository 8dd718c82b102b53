import json, os, sys, torch
sys.path.insert(0, os.getcwd())
from gnnadvisor_osdi21_amd import _lib, graph
dev = torch.device("cuda:0")
g = graph.make_config_graph("reddit-like", device=dev)
N, E, D = g.num_nodes, g.nnz, 64
A = torch.randn(N, D, device=dev); X = torch.randn(N, D, device=dev); out = torch.empty(E, device=dev)
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
ps = 128
pp, p2n = [t.to(dev) for t in _lib.build_part(ps, g.row_pointers.cpu())]
for ld in (128, 64):
    Xl = _lib.empty_rows(N, D, ld, dev); Xl.copy_(X)
    for B in (8, 12, 16, 20, 24, 32):
        for G in (8, 16, 32):
            _lib.reset_tuning(); _lib.set_tuning(column_phases=B, groups_per_chunk=G)
            _lib.prepare_graph(g.column_index, pp, p2n, N, N, ps, [D])
            ms = timed(lambda: _lib.sddmm(A, Xl, g.column_index, pp, p2n, ps, out=out))
            _lib.release_graph(g.column_index)
            print(json.dumps(dict(ld=ld, phases=B, groups_per_chunk=G, ms=round(ms, 4), G_edges_s=round(E / ms / 1e6, 1))), flush=True)
