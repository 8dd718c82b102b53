"""How many sets per workgroup should the sweep kernel run when the rows per edge-balanced set are uneven?  The
hidden-locality Reddit-like graph (D = 64) in four orders -- planted, scrambled, the product's renumbering, the Rabbit
Order restatement (degrees clustered: a tenth of the edges belong to rows beyond a set's 512 LDS accumulator rows at
2 sets per workgroup) -- prepared, kernel ms of: the library's own choice, the sweep kernel forced at R = 2, 3, 4, 6, 8
sets per workgroup (16 phases), and the streaming kernel.  Permutations from tools/_perm/ (tools/probe_reorder_quality.py).
usage: probe_sets.py [config] [orders]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnadvisor_osdi21_amd import _lib, graph  # noqa: E402

dev = torch.device("cuda:0")
cfg = sys.argv[1] if len(sys.argv) > 1 else "reddit-like"
orders = (sys.argv[2] if len(sys.argv) > 2 else "scrambled,product,rabbit,planted").split(",")
D, ps = 64, int(os.environ.get("PROBE_PS", "128"))
RS = [int(v) for v in os.environ.get("PROBE_RS", "2,3,4,6,8").split(",") if v]
g0 = graph.make_config_graph(cfg, device="cpu", locality=0.9, scale=1.0)
n = g0.num_nodes
rows = torch.repeat_interleave(torch.arange(n), (g0.row_pointers[1:] - g0.row_pointers[:-1]).long())
cols = g0.column_index.long()
perm = torch.randperm(n, generator=torch.Generator().manual_seed(1))
src, dst = perm[rows], perm[cols]
perm_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_perm")


def edges_of(order):
    if order == "planted":
        return rows, cols
    if order == "scrambled":
        return src, dst
    z = np.load(os.path.join(perm_dir, f"{cfg}_scale1.0_loc0.9_{order}.npz"))
    new_id = torch.from_numpy(z["new_id"]).long()
    return new_id[src], new_id[dst]


for order in orders:
    s, d = edges_of(order)
    g = graph.graph_from_edges(s.to(dev), d.to(dev), n)
    pp, p2n = _lib.build_part(ps, g.row_pointers.cpu())
    ppd, p2nd = pp.to(dev), p2n.to(dev)
    X = torch.randn(n, D, device=dev)
    out = torch.empty(n, D, device=dev)
    deg = (g.row_pointers[1:] - g.row_pointers[:-1]).to(torch.float32)

    def run(x=X, o=out):
        return _lib.sag(x, g.row_pointers, g.column_index, g.degrees, ppd, p2nd, ps, 32, 4, out=o)

    def timeit(k=10):
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        _lib.profile_begin(k)
        for _ in range(k):
            run()
        torch.cuda.synchronize()
        return round(_lib.profile_end()["main_ms"], 4)

    def exact():
        y = run(torch.ones_like(X), None)
        return bool((y == deg[:, None]).all())

    rec = dict(order=order, nnz=int(g.column_index.numel()))
    _lib.reset_tuning()
    _lib.set_tuning(pack_ids=1)
    before = _lib.runtime_counters()["sweep_launches"]
    rec["library_choice_ms"] = timeit()
    rec["library_choice"] = dict(phases=_lib.last_num_phases(), sweep=_lib.runtime_counters()["sweep_launches"] > before, exact=exact())
    for R in RS:
        _lib.reset_tuning()
        _lib.set_tuning(pack_ids=1, sweep=1, column_phases=16, groups_per_chunk=64 * R)
        rec[f"sweep_R{R}_ms"] = timeit()
        rec[f"sweep_R{R}_exact"] = exact()
    for B in (1, 2, 4, 8, 16):
        _lib.reset_tuning()
        _lib.set_tuning(pack_ids=1, sweep=2, column_phases=B)
        rec[f"stream_{B}_phases_ms"] = timeit()
    _lib.reset_tuning()
    _lib.release_graph(g.column_index)
    print(json.dumps(rec), flush=True)
    del g, X, out
    torch.cuda.empty_cache()
