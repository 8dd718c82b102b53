"""One configuration, a few steps: `GNNA_TUNE=... python tools/run_one.py [config] [D] [steps] [ps]` -- the workload of
PMC passes (tools/pmc_variants.sh) that compare kernel variants selected through GNNA_TUNE.  Prints kernel ms."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnadvisor_osdi21_amd import _lib, graph  # noqa: E402

dev = torch.device("cuda:0")
cfg = sys.argv[1] if len(sys.argv) > 1 else "reddit-like"
D = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ps = int(sys.argv[4]) if len(sys.argv) > 4 else 64
g = graph.make_config_graph(cfg, device=dev)
pp, p2n = _lib.build_part(ps, g.row_pointers.cpu())
ppd, p2nd = pp.to(dev), p2n.to(dev)
X = torch.randn(g.num_nodes, D, device=dev)
out = torch.empty_like(X)
_lib.sag(X, g.row_pointers, g.column_index, g.degrees, ppd, p2nd, ps, 32, 4, out=out)
torch.cuda.synchronize()
_lib.profile_begin(steps)
for _ in range(steps):
    _lib.sag(X, g.row_pointers, g.column_index, g.degrees, ppd, p2nd, ps, 32, 4, out=out)
torch.cuda.synchronize()
pr = _lib.profile_end()
print(json.dumps(dict(tune=os.environ.get("GNNA_TUNE", ""), cfg=cfg, D=D, ps=ps, kernel_ms=round(pr["main_ms"], 4),
                      phases=_lib.last_num_phases(), sweep_launches=_lib.runtime_counters()["sweep_launches"])))
