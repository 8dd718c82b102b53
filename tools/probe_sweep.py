"""Probe: the destination-blocked sweep kernel (gnna_sweep.hip) against the streaming kernel's sliced schedule.
Kernel ms (HIP events) on one graph / width for
  * the streaming kernel at its own choice and at forced phase counts,
  * the sweep kernel over phase counts x barrier slack (1 = strict, 1000 = none) x sets per workgroup R (0 = automatic).
usage: probe_sweep.py [config] [D] [phases,..] [slack,..] [R,..] [U]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnadvisor_osdi21_amd import _lib, graph  # noqa: E402

dev = torch.device("cuda:0")
cfg = sys.argv[1] if len(sys.argv) > 1 else "reddit-like"
D = int(sys.argv[2]) if len(sys.argv) > 2 else 64
phases = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "8,16,32").split(",")]
slacks = [int(v) for v in (sys.argv[4] if len(sys.argv) > 4 else "1,2,3,1000").split(",")]
Ks = [int(v) for v in (sys.argv[5] if len(sys.argv) > 5 else "0").split(",")]
U = int(sys.argv[6]) if len(sys.argv) > 6 else 4
ps = int(os.environ.get("PROBE_PS", "64"))
g = graph.make_config_graph(cfg, device=dev)
pp, p2n = _lib.build_part(ps, g.row_pointers.cpu())
ppd, p2nd = pp.to(dev), p2n.to(dev)
out = torch.empty(g.num_nodes, D, device=dev)
X = torch.randn(g.num_nodes, D, device=dev)
deg = (g.row_pointers[1:] - g.row_pointers[:-1]).to(torch.float32)


def run():
    return _lib.sag(X, g.row_pointers, g.column_index, g.degrees, ppd, p2nd, ps, 32, 4, out=out)


def timeit(n=10):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    _lib.profile_begin(n)
    for _ in range(n):
        run()
    torch.cuda.synchronize()
    pr = _lib.profile_end()
    return round(pr["main_ms"], 4), round(pr["prologue_ms"], 4)


def exact():
    ones = torch.ones_like(X)
    y = _lib.sag(ones, g.row_pointers, g.column_index, g.degrees, ppd, p2nd, ps, 32, 4)
    return bool((y == deg[:, None]).all())


_lib.reset_tuning()
_lib.set_tuning(loads_in_flight=U)
base = {"auto": timeit() + (_lib.last_num_phases(),)}
for B in phases:
    _lib.set_tuning(column_phases=B)
    base[B] = timeit()
print(json.dumps(dict(kernel="stream", cfg=cfg, D=D, ps=ps, U=U, ms_main_prologue=base)), flush=True)
for K in Ks:
    for sl in slacks:
        res = {}
        for B in phases:
            _lib.reset_tuning()
            kw = dict(loads_in_flight=U, sweep=1, sweep_slack=sl, column_phases=B,
                      blocks_per_cu=int(os.environ.get("PROBE_WGS", "0")), xcd_remap=int(os.environ.get("PROBE_DYN", "1")))
            if K:
                kw["groups_per_chunk"] = 64 * K
            _lib.set_tuning(**kw)
            before = _lib.runtime_counters()["sweep_launches"]
            res[B] = timeit() + (exact(),)
            assert _lib.runtime_counters()["sweep_launches"] > before
        print(json.dumps(dict(kernel="sweep", cfg=cfg, D=D, ps=ps, U=U, wgs=os.environ.get("PROBE_WGS", "0"), dynamic=os.environ.get("PROBE_DYN", "1"), R=K, slack=sl, ms_main_prologue_exact=res)), flush=True)
_lib.reset_tuning()
