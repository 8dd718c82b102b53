"""Probe: what calibrate_phases measures on a config graph (verbose timings per candidate phase count)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnadvisor_osdi21_amd import _lib, graph  # noqa: E402
from gnnadvisor_osdi21_amd.decider import calibrate_phases  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "papers100M-like"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 0.125
D = int(sys.argv[3]) if len(sys.argv) > 3 else 128
ps = int(sys.argv[4]) if len(sys.argv) > 4 else 16
g = graph.make_config_graph(cfg, device="cuda", scale=scale)
pp, p2n = _lib.build_part(ps, g.row_pointers.cpu())
for _ in range(3):
    print(calibrate_phases(g.column_index, pp.cuda(), p2n.cuda(), g.num_nodes, ps, [D], verbose=True))
