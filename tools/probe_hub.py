"""Hot-row cache variant of the streaming kernel against the plain packed path: `python tools/probe_hub.py [config] [D] [phases,..]`
(experiment switches through the environment: GNNA_HUB_REPS, GNNA_HUB_U4, GNNA_HUB_NOCACHE).  Prints kernel ms."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnadvisor_osdi21_amd import _lib, graph  # noqa: E402

dev = torch.device("cuda:0")
cfg = sys.argv[1] if len(sys.argv) > 1 else "reddit-like"
D = int(sys.argv[2]) if len(sys.argv) > 2 else 16
phases = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "0").split(",")]
g = graph.make_config_graph(cfg, device=dev)
avg = g.nnz / g.num_nodes
ps = 64 if avg >= 48 else (32 if avg >= 24 else 16)
pp, p2n = _lib.build_part(ps, g.row_pointers.cpu())
ppd, p2nd = pp.to(dev), p2n.to(dev)
X = torch.randn(g.num_nodes, D, device=dev)
out = torch.empty_like(X)


def kernel_ms(steps=20):
    call = lambda: _lib.sag(X, g.row_pointers, g.column_index, g.degrees, ppd, p2nd, ps, 32, 4, out=out)
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    _lib.profile_begin(steps)
    for _ in range(steps):
        call()
    torch.cuda.synchronize()
    pr = _lib.profile_end()
    return round(pr["main_ms"], 4)


env = {k: v for k, v in os.environ.items() if k.startswith("GNNA_HUB")}
for B in phases:
    row = dict(cfg=cfg, D=D, forced_phases=B, env=env)
    for name, kw in (("plain_packed", dict(row_cache=2)), ("hot_row_cache", dict(row_cache=0))):
        _lib.reset_tuning()
        _lib.set_tuning(column_phases=B, sweep=2, **kw)
        _lib.prepare_graph(g.column_index, ppd, p2nd, g.num_nodes, g.num_nodes, ps, [D])
        h0 = _lib.runtime_counters()["hub_launches"]
        row[name] = dict(kernel_ms=kernel_ms(), phases=_lib.last_num_phases(), hub=_lib.runtime_counters()["hub_launches"] > h0)
        _lib.release_graph(g.column_index)
    y_ref = None
    print(json.dumps(row), flush=True)
_lib.reset_tuning()
