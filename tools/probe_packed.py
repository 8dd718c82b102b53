"""Probe: packed column ids of a prepared graph (gnna_tuning.pack_ids) against reading the ids from column_index,
kernel ms by HIP events over phase counts.  usage: probe_packed.py [config] [D] [phases,..]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnadvisor_osdi21_amd import _lib, graph  # noqa: E402

dev = torch.device("cuda:0")
cfg = sys.argv[1] if len(sys.argv) > 1 else "reddit-like"
D = int(sys.argv[2]) if len(sys.argv) > 2 else 64
phases = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "0,4,6,8,10,12,16,24").split(",")]
ps = int(os.environ.get("PROBE_PS", "64"))
g = graph.make_config_graph(cfg, device=dev)
pp, p2n = _lib.build_part(ps, g.row_pointers.cpu())
ppd, p2nd = pp.to(dev), p2n.to(dev)
out = torch.empty(g.num_nodes, D, device=dev)
X = torch.randn(g.num_nodes, D, device=dev)
deg = (g.row_pointers[1:] - g.row_pointers[:-1]).to(torch.float32)


def timeit(n=20):
    for _ in range(3):
        _lib.sag(X, g.row_pointers, g.column_index, g.degrees, ppd, p2nd, ps, 32, 4, out=out)
    torch.cuda.synchronize()
    _lib.profile_begin(n)
    for _ in range(n):
        _lib.sag(X, g.row_pointers, g.column_index, g.degrees, ppd, p2nd, ps, 32, 4, out=out)
    torch.cuda.synchronize()
    return round(_lib.profile_end()["main_ms"], 4)


def exact():
    y = _lib.sag(torch.ones_like(X), g.row_pointers, g.column_index, g.degrees, ppd, p2nd, ps, 32, 4)
    return bool((y == deg[:, None]).all())


for B in phases:
    row = dict(cfg=cfg, D=D, B=B)
    sweep = int(os.environ.get("PROBE_SWEEP", "0"))      # 1: the sweep kernel instead of the streaming kernel
    for name, pk in (("ids_from_column_index", 2), ("packed_ids", 1)):
        _lib.reset_tuning()
        _lib.set_tuning(column_phases=B, pack_ids=pk, sweep=sweep, sweep_slack=int(os.environ.get("PROBE_SLACK", "0")),
                        blocks_per_cu=int(os.environ.get("PROBE_WGS", "0")), loads_in_flight=int(os.environ.get("PROBE_U", "-1")),
                        groups_per_chunk=(64 * int(os.environ["PROBE_ROUNDS"]) if "PROBE_ROUNDS" in os.environ else -1))
        _lib.prepare_graph(g.column_index, ppd, p2nd, g.num_nodes, g.num_nodes, ps, [D])
        before = _lib.runtime_counters()["packed_launches"]
        row[name] = timeit()
        row[name + "_exact"] = exact()
        row[name + "_packed_launches"] = _lib.runtime_counters()["packed_launches"] - before
        row["phases"] = _lib.last_num_phases()
    print(json.dumps(row), flush=True)
_lib.reset_tuning()
