"""Would the real kernel gain from rows laid out clear of the slow line class (profiles/r4/gather_floor_row_positions.log)?
Emulated from outside the library: the source matrix gets dummy rows at the slots to avoid and the column ids are renumbered
to slots (a rectangular call), everything else as in production (prepared graph, packed ids).  D = 16 / 8 / 4.
usage: python tools/probe_slotted_rows.py [config] [ps]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnadvisor_osdi21_amd import _lib, graph  # noqa: E402

dev = torch.device("cuda:0")
cfg = sys.argv[1] if len(sys.argv) > 1 else "reddit-like"
ps = int(sys.argv[2]) if len(sys.argv) > 2 else 128
g = graph.make_config_graph(cfg, device=dev)
pp, p2n = _lib.build_part(ps, g.row_pointers.cpu())
ppd, p2nd = pp.to(dev), p2n.to(dev)
N, E = g.num_nodes, int(g.column_index.numel())


def run(tag, X, col, n_in, D):
    _lib.prepare_graph(col, ppd, p2nd, n_in, N, ps, [D])
    out = torch.empty(N, D, device=dev)
    steps = 20
    for _ in range(3):
        _lib.agg_rect(0, X, col, ppd, p2nd, N, ps, out=out)
    torch.cuda.synchronize()
    _lib.profile_begin(steps)
    for _ in range(steps):
        _lib.agg_rect(0, X, col, ppd, p2nd, N, ps, out=out)
    torch.cuda.synchronize()
    pr = _lib.profile_end()
    print(json.dumps(dict(cfg=cfg, D=D, layout=tag, rows=n_in, kernel_ms=round(pr["main_ms"], 4), phases=_lib.last_num_phases(),
                          packed=_lib.runtime_counters()["packed_launches"])), flush=True)
    _lib.release_graph(col)
    return out


for D in (16, 8, 4):
    X = torch.randn(N, D, device=dev)
    ref = run("contiguous", X, g.column_index, N, D)
    s = 1024 // (4 * D)                  # row slots per KiB
    for name, skip in (("bytes 384..511 of every 1024 empty", [(3 * s // 8, s // 8)]),
                       ("bytes 384..511 of every 512 empty", [(3 * s // 8, s // 8), (7 * s // 8, s // 8)])):
        usable = torch.ones(s, dtype=torch.bool)
        for a, n in skip:
            usable[a:a + n] = False
        pos = torch.nonzero(usable).flatten().to(dev)          # usable slot positions inside a KiB
        u = int(pos.numel())
        ids = torch.arange(N, device=dev)
        slot = (ids // u) * s + pos[ids % u]
        n_in = ((N + u - 1) // u) * s
        Xs = torch.zeros(n_in, D, device=dev)
        Xs[slot] = X
        col2 = slot[g.column_index.long()].int().contiguous()
        out = run(name, Xs, col2, n_in, D)
        assert torch.allclose(out, ref, rtol=1e-4, atol=1e-3), (out - ref).abs().max()
