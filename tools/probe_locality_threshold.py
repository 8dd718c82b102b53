"""Where does the sweep kernel stop paying as the column ids get local?  Reddit-like graphs (D = 64, partSize 128) generated
with a share `locality` of their edges within +-4,096 ids of the destination (planted order, not scrambled): kernel ms of the
sweep kernel forced (16 phases), the streaming kernel at its own phase count and single pass, and of the library's own
choice -- beside the statistic the choice goes by (share of the edges within 1/16 of the rows of their destination).
usage: probe_locality_threshold.py [localities]      (PROBE_D = row width, default 64)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnadvisor_osdi21_amd import _lib, graph  # noqa: E402

dev = torch.device("cuda:0")
locs = [float(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,0.15,0.25,0.35,0.5,0.65,0.8").split(",")]
D, ps = int(os.environ.get("PROBE_D", "64")), 128
for loc in locs:
    g = graph.make_config_graph("reddit-like", device=dev, locality=loc)
    n = g.num_nodes
    rows = torch.repeat_interleave(torch.arange(n, device=dev), (g.row_pointers[1:] - g.row_pointers[:-1]).long())
    share = float(((rows - g.column_index.long()).abs() < n / 16).float().mean())
    del rows
    pp, p2n = _lib.build_part(ps, g.row_pointers.cpu())
    ppd, p2nd = pp.to(dev), p2n.to(dev)
    X = torch.randn(n, D, device=dev)
    out = torch.empty(n, D, device=dev)

    def timeit(k=10):
        fn = lambda: _lib.sag(X, g.row_pointers, g.column_index, g.degrees, ppd, p2nd, ps, 32, 4, out=out)  # noqa: E731
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        _lib.profile_begin(k)
        for _ in range(k):
            fn()
        torch.cuda.synchronize()
        return round(_lib.profile_end()["main_ms"], 4)

    rec = dict(D=D, locality=loc, nnz=int(g.column_index.numel()), share_within_a_sixteenth_of_the_rows=round(share, 3))
    _lib.reset_tuning(); _lib.set_tuning(pack_ids=1)
    before = _lib.runtime_counters()["sweep_launches"]
    rec["library_choice_ms"] = timeit()
    rec["library_choice"] = dict(phases=_lib.last_num_phases(), sweep=_lib.runtime_counters()["sweep_launches"] > before)
    _lib.reset_tuning(); _lib.set_tuning(pack_ids=1, sweep=1, column_phases=16)
    rec["sweep_16_phases_ms"] = timeit()
    _lib.reset_tuning(); _lib.set_tuning(pack_ids=1, sweep=2)
    rec["stream_own_phases_ms"] = timeit()
    rec["stream_own_phases"] = _lib.last_num_phases()
    for B in (1, 4, 8, 16):
        _lib.reset_tuning(); _lib.set_tuning(pack_ids=1, sweep=2, column_phases=B)
        rec[f"stream_{B}_ms"] = timeit()
    _lib.reset_tuning()
    _lib.release_graph(g.column_index)
    print(json.dumps(rec), flush=True)
    del g, X, out
    torch.cuda.empty_cache()
