"""SDDMM ms over row widths: `[GNNA_TUNE=U=8] python tools/probe_sddmm.py [config] [ps] [dims]` (edge_out[e] = <A[row(e)], B[col(e)]>)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnadvisor_osdi21_amd import _lib, graph  # noqa: E402

dev = torch.device("cuda:0")
cfg = sys.argv[1] if len(sys.argv) > 1 else "reddit-like"
ps = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dims = [int(d) for d in sys.argv[3].split(",")] if len(sys.argv) > 3 else [16, 32, 64, 128]
g = graph.make_config_graph(cfg, device=dev)
pp, p2n = _lib.build_part(ps, g.row_pointers.cpu())
ppd, p2nd = pp.to(dev), p2n.to(dev)
N, E = g.num_nodes, int(g.column_index.numel())
for D in dims:
    A = torch.randn(N, D, device=dev)
    ld = int(os.environ.get("SDDMM_LD", "0")) or D
    Bm = _lib.empty_rows(N, D, ld, dev)
    Bm.copy_(torch.randn(N, D, device=dev))
    out = torch.empty(E, device=dev)
    for _ in range(3):
        _lib.sddmm(A, Bm, g.column_index, ppd, p2nd, ps, out=out)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(10):
            _lib.sddmm(A, Bm, g.column_index, ppd, p2nd, ps, out=out)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 100)
    print(json.dumps(dict(tune=os.environ.get("GNNA_TUNE", ""), cfg=cfg, D=D, ld_src=ld, ps=ps, ms=round(best, 4), phases=_lib.last_num_phases(),
                          G_edges_s=round(E / best / 1e6, 1))), flush=True)
