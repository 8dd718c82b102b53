"""SDDMM ms on the headline graph, prepared (packed ids), partSize 128, contiguous and gapped source layouts -- run under
GNNA_LIB=<an ablation build> (tools/ceiling/build_sddmm_r6_ablations.sh); results are WRONG by construction there."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnadvisor_osdi21_amd import _lib, graph
dev = torch.device("cuda:0")
g = graph.make_config_graph("reddit-like", device=dev)
N, E, D, ps = g.num_nodes, g.nnz, 64, 128
A = torch.randn(N, D, device=dev); X = torch.randn(N, D, device=dev); out = torch.empty(E, device=dev)
pp, p2n = [t.to(dev) for t in _lib.build_part(ps, g.row_pointers.cpu())]
_lib.set_graph_hints(g.column_index, 492, True)
_lib.prepare_graph(g.column_index, pp, p2n, N, N, ps, [D])
def timed(fn, reps=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for ld in (64, 128):
    Xl = _lib.empty_rows(N, D, ld, dev); Xl.copy_(X)
    ms = timed(lambda: _lib.sddmm(A, Xl, g.column_index, pp, p2n, ps, out=out))
    print(json.dumps(dict(lib=os.path.basename(os.environ.get("GNNA_LIB", "libgnna.so")), ld_src=ld, ms=round(ms, 4), G_edges_s=round(E / ms / 1e6, 1))), flush=True)
