"""Probe: very-low-degree graphs (reference Type II / III-like shapes) at D = 16 / 64."""
import sys, os, torch, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnadvisor_osdi21_amd import _lib, graph
dev = torch.device("cuda:0")
for (n, e, name) in ((1_900_000, 4_000_000, "typeII-like avg2"), (1_000_000, 8_000_000, "avg8"), (550_000, 1_850_000, "com-amazon-like")):
    g = graph.uniform_graph(n, e, seed=1, device=dev)
    for ps in (16, 32):
        pp, p2n = _lib.build_part(ps, g.row_pointers.cpu())
        ppd, p2nd = pp.to(dev), p2n.to(dev)
        for D in (16, 64):
            X = torch.randn(n, D, device=dev); out = torch.empty_like(X)
            for G in (16,):
                _lib.set_tuning(groups_per_chunk=G)
                fn = lambda: _lib.sag(X, g.row_pointers, g.column_index, g.degrees, ppd, p2nd, ps, 32, 4, out=out)
                for _ in range(3): fn()
                torch.cuda.synchronize(); _lib.profile_begin(20)
                for _ in range(20): fn()
                torch.cuda.synchronize(); r = _lib.profile_end()
                b = g.nnz * (4 * D + 4) + n * (4 * D + 4) + p2n.numel() * 8
                print(name, "nnz", g.nnz, "ps", ps, "D", D, "G", G, "ms", round(r["main_ms"], 4), "pro", round(r["prologue_ms"], 4),
                      "Gedges", round(g.nnz / r["main_ms"] / 1e6, 1), "TB/s", round(b / r["main_ms"] / 1e9, 2), flush=True)
