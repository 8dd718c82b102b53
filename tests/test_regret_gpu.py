"""Regret of the library's own schedule choice on graphs its rules were NOT fitted on.

`choose_slices`, `sweep_auto_phases` and `pick_row_stride` (csrc/gnna_agg.hip) hold about fifteen literal thresholds, all
measured on Chung-Lu power-law graphs with random ids.  Here: R-MAT (a, b, c = 0.57, 0.19, 0.19; permuted and
in its own structured numbering), a community-structured graph (ordered and scrambled), plus the fitted families at
widths between the fitted ones.  For every (graph, D) the automatic choice is timed against forced schedules --
the streaming kernel at 1 / 4 / 8 / 16 / 32 phases and the sweep kernel at 8 / 16 -- and must be within 7 % of the
best forced one.  The table goes to gpurun_out/ (and from there to profiles/)."""
import os
import time

import pytest
import torch

from gnnadvisor_osdi21_amd import _lib, graph
from gnnadvisor_osdi21_amd.decider import choose_part_size

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _time_ms(fn, reps=12, rounds=3):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(rounds):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3 / reps)
    return best


CASES = [
    ("rmat-2^18-permuted", lambda: graph.rmat_graph(1 << 18, 70_000_000, seed=11, device="cuda"), (16, 64, 128)),
    ("rmat-2^18-own-order", lambda: graph.rmat_graph(1 << 18, 70_000_000, seed=11, device="cuda", permute=False), (64,)),
    ("rmat-2^21-permuted", lambda: graph.rmat_graph(1 << 21, 80_000_000, seed=12, device="cuda"), (64,)),
    ("community-300k-ordered", lambda: graph.community_graph(300_000, 60_000_000, 40, seed=13, device="cuda"), (64,)),
    ("community-300k-scrambled", lambda: graph.community_graph(300_000, 60_000_000, 40, seed=13, device="cuda", scramble=True), (32, 64)),
    ("reddit-like", lambda: graph.make_config_graph("reddit-like", device="cuda"), (41, 100)),
    ("products-like-half", lambda: graph.make_config_graph("products-like", device="cuda", scale=0.5), (64,)),
    ("amazon0505-like", lambda: graph.make_config_graph("amazon0505-like", device="cuda"), (64,)),
    # round 5: long rows whose ids are partly local (what a community order looks like to the kernel): the sweep kernel's
    # lock-step walk loses there, and single pass only wins once ~3/4 of the edges sit inside an L2-sized window
    ("reddit-like-half-local", lambda: graph.make_config_graph("reddit-like", device="cuda", locality=0.5), (64,)),
    ("reddit-like-two-thirds-local", lambda: graph.make_config_graph("reddit-like", device="cuda", locality=0.65), (64, 48, 32, 16)),
]
FORCED = [("stream-1", dict(column_phases=1, sweep=2)), ("stream-4", dict(column_phases=4, sweep=2)),
          ("stream-8", dict(column_phases=8, sweep=2)), ("stream-16", dict(column_phases=16, sweep=2)),
          ("stream-32", dict(column_phases=32, sweep=2)), ("sweep-8", dict(column_phases=8, sweep=1)),
          ("sweep-16", dict(column_phases=16, sweep=1))]


def test_the_automatic_schedule_is_within_7_percent_of_the_best_forced_one():
    t = _lib.get_tuning()
    if t["column_phases"] != 0 or t["sweep"] != 0:
        pytest.skip("GNNA_TUNE forces a schedule: the automatic choice is not under test")
    lines, worst = [], (0.0, None)
    for name, make, dims in CASES:
        g = make()
        avg = g.nnz / g.num_nodes
        ps = choose_part_size(avg, 64)                                # the mi355x policy
        pp, p2n = _lib.build_part(ps, g.row_pointers.cpu())
        ppd, p2nd = pp.cuda(), p2n.cuda()
        for D in dims:
            X = torch.randn(g.num_nodes, D, device="cuda", generator=torch.Generator(device="cuda").manual_seed(D))
            out = torch.empty_like(X)
            run = lambda: _lib.sag(X, g.row_pointers, g.column_index, g.degrees, ppd, p2nd, ps, 32, 4, out=out)
            try:
                _lib.set_tuning(pack_ids=2)                            # ids from column_index everywhere: like for like
                auto_ms = _time_ms(run)
                auto_phases = _lib.last_num_phases()
                swept0 = _lib.runtime_counters()["sweep_launches"]
                run()
                auto_kernel = "sweep" if _lib.runtime_counters()["sweep_launches"] > swept0 else "stream"
                forced = {}
                for tag, kw in FORCED:
                    if tag.startswith("sweep") and D > 128:
                        continue
                    _lib.set_tuning(pack_ids=2, **kw)
                    forced[tag] = _time_ms(run)
            finally:
                _lib.reset_tuning()
            best_tag = min(forced, key=forced.get)
            regret = auto_ms / forced[best_tag] - 1.0
            if regret > 0.05:
                # a timing test must not fail on one noisy sample: measure the two contenders again, longer
                try:
                    _lib.set_tuning(pack_ids=2)
                    auto_ms = min(auto_ms, _time_ms(run, reps=30, rounds=4))
                    _lib.set_tuning(pack_ids=2, **dict(FORCED)[best_tag])
                    forced[best_tag] = min(forced[best_tag], _time_ms(run, reps=30, rounds=4))
                finally:
                    _lib.reset_tuning()
                regret = auto_ms / forced[best_tag] - 1.0
            lines.append(f"{name:28s} N={g.num_nodes:8d} nnz={g.nnz:10d} D={D:3d} ps={ps:2d} | auto {auto_kernel}-{auto_phases:<2d} "
                         f"{auto_ms:7.3f} ms | best {best_tag:9s} {forced[best_tag]:7.3f} ms | regret {100 * regret:+5.1f} % | "
                         + " ".join(f"{k}={v:.3f}" for k, v in forced.items()))
            if regret > worst[0]:
                worst = (regret, lines[-1])
            del X, out
        _lib.release_graph(g.column_index)
        del g, ppd, p2nd
        torch.cuda.empty_cache()
    table = "\n".join(lines)
    print("\n" + table)
    outdir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(outdir):
        with open(os.path.join(outdir, "regret_table.log"), "w") as f:
            f.write("# automatic schedule vs forced schedules, kernel + prologue wall ms per call (best of 3 x 12 calls), ids from column_index\n")
            f.write(table + "\n")
    assert worst[0] <= 0.07, "the automatic choice loses more than 7 % somewhere:\n" + str(worst[1])
