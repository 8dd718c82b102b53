"""Which rows differ between zero_fill=1 and zero_fill=2 on a mid-sized graph (debug aid)."""
import numpy as np
import torch
from gnnadvisor_osdi21_amd import _lib, graph

for n, e, D, ps in ((300000, 4500000, 128, 16), (232965, 20000000, 64, 64)):
    g = graph.powerlaw_graph(n, e, 2000, seed=3)
    pp, p2n = _lib.build_part(ps, g.row_pointers)
    X = torch.randn(n, D, device="cuda")
    rp, ci, deg, ppd, p2nd = (t.cuda() for t in (g.row_pointers, g.column_index, g.degrees, pp, p2n))
    for G in (16, 64):
        _lib.set_tuning(groups_per_chunk=G, zero_fill=2, column_phases=1)
        ref = _lib.sag(X, rp, ci, deg, ppd, p2nd, ps, 32, 4).clone()
        _lib.set_tuning(zero_fill=1)
        out = torch.full((n, D), float("nan"), device="cuda")
        _lib.sag(X, rp, ci, deg, ppd, p2nd, ps, 32, 4, out=out)
        torch.cuda.synchronize()
        bad = ~((out == ref).all(dim=1))
        idx = bad.nonzero().flatten().cpu().numpy()
        d = (g.row_pointers[1:] - g.row_pointers[:-1]).numpy()
        print(f"n={n} D={D} ps={ps} G={G}: {len(idx)} rows differ; degrees of the first: {d[idx[:12]]}, rows {idx[:12]}")
        if len(idx):
            r = int(idx[0]); p2 = p2n.numpy()
            gi = np.searchsorted(p2, r)
            print("  first group of row", gi, "gi % G", gi % G, "nan?", bool(torch.isnan(out[r]).any()), out[r, :4].cpu().numpy(), ref[r, :4].cpu().numpy())
    _lib.reset_tuning()
