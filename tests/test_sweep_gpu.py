"""GPU parity of the destination-blocked sweep kernel (csrc/gnna_sweep.hip) -- the sliced schedule with the
partial rows kept in LDS across the source slices -- against the CPU oracle, through the C ABI.

Same bars as test_parity_gpu.py: X = ones exact (reference unitest.py:27,54-63), random inputs within
1e-4 * max(1, scale) of the fp64 CSR formula.  Every case also checks that the sweep kernel really ran
(gnna_runtime_counters), so a silent fall-back to the streaming kernel cannot pass.
"""
import numpy as np
import pytest
import torch

import oracle
from gnnadvisor_osdi21_amd import _lib, graph
from util import assert_close_f64, dev, make_case

pytestmark = pytest.mark.gpu


class sweep_forced:
    """Process-wide knobs for one case: sweep kernel on, `phases` forced, optional sets per workgroup and slack."""

    def __init__(self, phases, rounds=0, slack=0, wgs=0, dynamic=1, **kw):
        # wgs: workgroups per CU (1: 16 wavefronts with all of the CU's LDS, 2: 32 wavefronts, half the rows each)
        # (wide_blocks = 2: one sweep launch per call -- these tests count launches)
        self.kw = dict(column_phases=phases, sweep=1, sweep_slack=slack, blocks_per_cu=wgs, deterministic=0, xcd_remap=dynamic,
                       wide_blocks=2, pack_ids=0, **kw)
        if rounds:
            self.kw["groups_per_chunk"] = 64 * rounds

    def __enter__(self):
        _lib.reset_tuning()
        _lib.set_tuning(**self.kw)
        self.before = _lib.runtime_counters()["sweep_launches"]
        return self

    def launches(self):
        return _lib.runtime_counters()["sweep_launches"] - self.before

    def __exit__(self, *exc):
        _lib.reset_tuning()


def check_modes(g, X, pp, p2n, ps, eps=0.5, what="", sum_scale=False):
    Xd, rp, ci, deg, ppd, p2nd = dev(X, g.row_pointers, g.column_index, g.degrees, pp, p2n)
    ys = _lib.sag(Xd, rp, ci, deg, ppd, p2nd, ps, 32, 4)
    yg = _lib.agg_gcn(Xd, rp, ci, deg, ppd, p2nd, ps, 32, 4)
    yi = _lib.agg_gin(Xd, rp, ci, eps, ppd, p2nd, ps, 32, 4)
    torch.cuda.synchronize()
    Xn, cin, rpn, degn = X.numpy(), g.column_index.numpy(), g.row_pointers.numpy(), g.degrees.numpy()
    # SAG / GIN: the strict form of SURVEY appendix A, |err| <= 1e-4 * max(1, |ref|), wherever no row has more than 4096
    # edges.  Longer rows are held to 1e-4 of the sum of |terms| (what fp32 summation error scales with): a sum of 8,000
    # N(0, 1) terms is ~90 in magnitude, every fp32 add at that magnitude rounds by ~5e-6, and over 10 M output elements one
    # of them lands at 1.5e-4 while its own |ref| is < 1 (measured, round 5) -- no uncompensated fp32 summation, the
    # reference's included, meets the strict form there.  GCN-weighted sums always use the sum of |terms|
    sscale = oracle.csr_f64(0, np.abs(Xn), rpn, cin) if (sum_scale and int(np.diff(rpn).max()) > 4096) else None
    assert_close_f64(ys.cpu().numpy(), oracle.csr_f64(0, Xn, rpn, cin), what=what + " sag vs fp64", scale=sscale)
    gscale = oracle.csr_f64(1, np.abs(Xn), rpn, cin, degn)
    assert_close_f64(yg.cpu().numpy(), oracle.csr_f64(1, Xn, rpn, cin, degn), what=what + " gcn vs fp64", scale=gscale)
    assert_close_f64(yi.cpu().numpy(), oracle.csr_f64(2, Xn, rpn, cin, None, eps), what=what + " gin vs fp64", scale=sscale)
    assert_close_f64(ys.cpu().numpy(), oracle.sag(Xn, cin, pp.numpy(), p2n.numpy()), what=what + " sag vs oracle", scale=sscale)


@pytest.mark.parametrize("dim", [4, 6, 7, 16, 22, 32, 41, 47, 64, 100, 128])
@pytest.mark.parametrize("phases,wgs,dynamic", [(2, 1, 1), (5, 2, 1), (32, 1, 0), (32, 2, 1), (8, 1, 0)])
def test_sweep_matches_oracle_over_widths_and_phase_counts(dim, phases, wgs, dynamic):
    g, X, pp, p2n = make_case(3000, 200000, dim, 16, seed=dim * 7 + phases, kind="powerlaw")
    with sweep_forced(phases, wgs=wgs, dynamic=dynamic, gcn_prescale=1) as s:
        check_modes(g, X, pp, p2n, 16, what=f"sweep dim={dim} phases={phases}")
        assert s.launches() == 3, "the sweep kernel did not run"
        assert _lib.last_num_phases() == phases


@pytest.mark.parametrize("ps", [1, 3, 8, 32, 64, 100])
@pytest.mark.parametrize("rounds", [1, 2, 5])
def test_sweep_part_sizes_and_sets_per_workgroup(ps, rounds):
    g, X, pp, p2n = make_case(2500, 150000, 64, ps, seed=ps + rounds, kind="powerlaw")
    with sweep_forced(8, rounds=rounds, gcn_prescale=1) as s:
        check_modes(g, X, pp, p2n, ps, what=f"sweep ps={ps} rounds={rounds}")
        assert s.launches() == 3


def test_sweep_ones_is_exact():
    for dim, n, e in ((64, 5000, 600000), (16, 4000, 300000), (128, 2000, 200000)):
        g, X, pp, p2n = make_case(n, e, dim, 32, seed=dim, kind="powerlaw", x="ones")
        Xd, rp, ci, deg, ppd, p2nd = dev(X, g.row_pointers, g.column_index, g.degrees, pp, p2n)
        for phases in (3, 16, 32):
            with sweep_forced(phases) as s:
                y = _lib.sag(Xd, rp, ci, deg, ppd, p2nd, 32, 32, 4).cpu().numpy()
                assert s.launches() == 1
            want = np.repeat((g.row_pointers[1:] - g.row_pointers[:-1]).numpy().astype(np.float32)[:, None], dim, 1)
            assert np.array_equal(y, want), (dim, phases)


@pytest.mark.parametrize("dynamic", [0, 1])
def test_sweep_long_rows_crossing_the_chunks_of_a_wavefront_share_are_exact(dynamic):
    """Part size 1 makes every edge a group: a wavefront's share is several 64-group chunks and most rows reach from
    an interior chunk of one share into the next share -- the rows two wavefronts add to concurrently.  (The static
    shares once marked only the first / last CHUNK's border rows as shared and lost updates on full-size graphs.)"""
    g, X, pp, p2n = make_case(3000, 1200000, 64, 1, seed=77, kind="powerlaw", x="ones")
    Xd, rp, ci, deg, ppd, p2nd = dev(X, g.row_pointers, g.column_index, g.degrees, pp, p2n)
    want = (g.row_pointers[1:] - g.row_pointers[:-1]).to(torch.float32)[:, None].expand(-1, 64)
    for phases, slack in ((6, 1000), (8, 2), (16, 1)):
        with sweep_forced(phases, slack=slack, dynamic=dynamic) as s:
            for rep in range(4):
                y = _lib.sag(Xd, rp, ci, deg, ppd, p2nd, 1, 32, 4).cpu()
                assert torch.equal(y, want), (dynamic, phases, slack, rep, int((y != want).any(1).sum()))
            assert s.launches() == 4


@pytest.mark.parametrize("dim,ps,phases", [(64, 32, 8), (16, 8, 3), (128, 64, 16), (41, 16, 32), (100, 3, 5)])
def test_sweep_reads_the_packed_ids_of_a_prepared_graph(dim, ps, phases):
    """A prepared graph's packed column ids (gnna_tuning.pack_ids) serve the sweep kernel too: its sets then start at
    multiples of 64 groups, so that a set's chunks are the chunks the copy is laid out by."""
    g, X, pp, p2n = make_case(4000, 400000, dim, ps, seed=dim + phases, kind="powerlaw")
    Xd, rp, ci, deg, ppd, p2nd = dev(X, g.row_pointers, g.column_index, g.degrees, pp, p2n)
    try:
        with sweep_forced(phases, gcn_prescale=1) as s:
            _lib.prepare_graph(ci, ppd, p2nd, g.num_nodes, g.num_nodes, ps, [dim])
            before = _lib.runtime_counters()["packed_launches"]
            check_modes(g, X, pp, p2n, ps, what=f"sweep packed dim={dim} ps={ps} phases={phases}")
            # check_modes moves its own copies of the index tensors to the device: run the prepared tensors as well
            ys = _lib.sag(Xd, rp, ci, deg, ppd, p2nd, ps, 32, 4)
            assert _lib.runtime_counters()["packed_launches"] == before + 1
            assert s.launches() == 4
        assert_close_f64(ys.cpu().numpy(), oracle.csr_f64(0, X.numpy(), g.row_pointers.numpy(), g.column_index.numpy()),
                         what="sweep packed sag")
        ones = torch.ones_like(Xd)
        with sweep_forced(phases):
            y1 = _lib.sag(ones, rp, ci, deg, ppd, p2nd, ps, 32, 4).cpu()
        want = (g.row_pointers[1:] - g.row_pointers[:-1]).to(torch.float32)[:, None].expand(-1, dim)
        assert torch.equal(y1, want)
    finally:
        _lib.reset_tuning()
        _lib.release_graph(ci)


def test_the_library_picks_the_sweep_kernel_only_where_it_wins():
    """gnna_tuning.sweep = 0: rows of 33-64 floats, a sliced schedule over a square problem with long rows (>= 300 edges)
    that fit the accumulators in two sets -> sweep_kernel; narrower rows, short rows, rectangular problems and sweep = 2
    -> stream_kernel.  Wide rows of such a graph run in 64-float column blocks (round 4), each of which is a sweep call;
    with the blocks switched off they stay on stream_kernel.  All give the oracle's result."""
    t = _lib.get_tuning()
    if t["column_phases"] != 0 or t["sweep"] != 0 or t["deterministic"] != 0 or t["wide_blocks"] != 0:
        pytest.skip("GNNA_TUNE forces the schedule: the automatic choice is not under test")
    g = graph.powerlaw_graph(80000, 32000000, 8000, seed=5, device="cuda")           # ~400 edges per row, X = 20 MB at D = 64
    n = g.num_nodes
    pp, p2n = _lib.build_part(64, g.row_pointers.cpu())
    ppd, p2nd = pp.cuda(), p2n.cuda()
    rpn, cin = g.row_pointers.cpu().numpy(), g.column_index.cpu().numpy()
    _lib.reset_tuning()

    def launches(fn):
        before = _lib.runtime_counters()["sweep_launches"]
        y = fn()
        torch.cuda.synchronize()
        return _lib.runtime_counters()["sweep_launches"] - before, y

    try:
        for dim, swept, blocks in ((64, 1, 0), (41, 1, 0), (32, 0, 0), (128, 2, 0), (128, 0, 2)):
            X = torch.randn(n, dim, generator=torch.Generator().manual_seed(dim))
            Xd = X.cuda()
            _lib.set_tuning(wide_blocks=blocks)
            k, y = launches(lambda: _lib.sag(Xd, g.row_pointers, g.column_index, g.degrees, ppd, p2nd, 64, 32, 4))
            assert k == swept, (dim, k, _lib.last_num_phases())
            assert _lib.last_num_phases() >= 2
            # (rows of up to 8,000 edges: 1e-4 of the sum of |terms|; the strict form measured 1 element of 10.24 M at 1.5e-4)
            assert_close_f64(y.cpu().numpy(), oracle.csr_f64(0, X.numpy(), rpn, cin), what=f"automatic choice, dim {dim}",
                             scale=oracle.csr_f64(0, np.abs(X.numpy()), rpn, cin))
        X = torch.randn(n, 64, generator=torch.Generator().manual_seed(1)).cuda()
        _lib.set_tuning(sweep=2)
        k, _ = launches(lambda: _lib.sag(X, g.row_pointers, g.column_index, g.degrees, ppd, p2nd, 64, 32, 4))
        assert k == 0
        _lib.reset_tuning()
        # a destination shard (rows != source rows) of the same graph keeps the streaming kernel
        k, _ = launches(lambda: _lib.agg_rect(0, X, g.column_index[: int(g.row_pointers[n // 2])].contiguous(),
                                              *[t.cuda() for t in _lib.build_part(64, g.row_pointers[: n // 2 + 1].cpu())], n // 2, 64))
        assert k == 0
    finally:
        _lib.reset_tuning()
        _lib.release_graph(None)
    # short rows (40 edges per row): streaming kernel
    g2 = graph.powerlaw_graph(200000, 8000000, 2000, seed=6, device="cuda")
    pp2, p2n2 = _lib.build_part(32, g2.row_pointers.cpu())
    X2 = torch.randn(g2.num_nodes, 64, device="cuda")
    k, _ = launches(lambda: _lib.sag(X2, g2.row_pointers, g2.column_index, g2.degrees, pp2.cuda(), p2n2.cuda(), 32, 32, 4))
    assert k == 0
    _lib.release_graph(None)


def test_the_library_keeps_locality_ordered_graphs_off_the_sweep_kernel():
    """Round 5: the lock-step walk of the sweep kernel pays only for SCATTERED ids.  A long-row graph with half of its
    edges within a window of the destination's id (what a community order -- Rabbit-ordered data, a renumbered graph -- looks
    like to the kernel) ran 1.72 ms on the sweep kernel against 1.42 on the streaming kernel's sliced schedule at Reddit size
    (profiles/r5/sweep_and_stream_over_locality.log; the Rabbit-ordered graph itself 1.94 against 1.34): sweep_auto_phases
    now reads the share of the edges within one sweep slice (1/16 of the rows) of their destination and stays out above 0.3.
    Such a graph still runs SLICED (its share inside an L2-sized window is below the single-pass threshold), on stream_kernel;
    the same graph with scrambled ids takes the sweep kernel as before.  Both give the exact known answer."""
    t = _lib.get_tuning()
    if t["column_phases"] != 0 or t["sweep"] != 0 or t["deterministic"] != 0 or t["wide_blocks"] != 0:
        pytest.skip("GNNA_TUNE forces the schedule: the automatic choice is not under test")
    n = 80000
    g = graph.powerlaw_graph(n, 32000000, 8000, seed=5, device="cuda", locality=0.5, window=2048)      # ~400 edges per row
    rows = torch.repeat_interleave(torch.arange(n, device="cuda"), (g.row_pointers[1:] - g.row_pointers[:-1]).long())
    share = float(((rows - g.column_index.long()).abs() < n / 16).float().mean())
    assert 0.45 < share < 0.7, share
    perm = torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    g_scr = graph.graph_from_edges(perm[rows], perm[g.column_index.long()], n)
    X = torch.ones(n, 64, device="cuda")
    _lib.reset_tuning()
    try:
        for name, gg, swept in (("half of the edges local", g, 0), ("scrambled", g_scr, 1)):
            pp, p2n = _lib.build_part(64, gg.row_pointers.cpu())
            ppd, p2nd = pp.cuda(), p2n.cuda()
            before = _lib.runtime_counters()["sweep_launches"]
            y = _lib.sag(X, gg.row_pointers, gg.column_index, gg.degrees, ppd, p2nd, 64, 32, 4)
            torch.cuda.synchronize()
            k = _lib.runtime_counters()["sweep_launches"] - before
            assert k == swept, (name, k, _lib.last_num_phases())
            assert _lib.last_num_phases() >= 2, (name, _lib.last_num_phases())
            deg = (gg.row_pointers[1:] - gg.row_pointers[:-1]).to(torch.float32)
            assert torch.equal(y, deg[:, None].expand(n, 64)), name
    finally:
        _lib.reset_tuning()
        _lib.release_graph(None)


def test_sweep_rows_beyond_the_accumulators_take_the_atomic_path():
    """Low-degree rows: a set (1/256 of the edges with one set per workgroup) spans more destination rows than the
    CU's LDS holds (256 rows of 128 floats, 512 of 64) -- the rows beyond are flushed per slice with atomics."""
    for dim, n, e in ((128, 150000, 500000), (64, 300000, 900000), (16, 600000, 1500000)):
        g, X, pp, p2n = make_case(n, e, dim, 4, seed=dim + 1)
        for rounds, wgs, dynamic in ((1, 1, 1), (3, 2, 0)):
            with sweep_forced(4, rounds=rounds, wgs=wgs, dynamic=dynamic, gcn_prescale=1) as s:
                check_modes(g, X, pp, p2n, 4, what=f"sweep overflow dim={dim} rounds={rounds}")
                assert s.launches() == 3


def test_sweep_hub_row_spanning_many_sets_and_rows_without_edges():
    n = 40000
    src = torch.cat([torch.zeros(n - 1, dtype=torch.int64), torch.arange(1, n)])
    dst = torch.cat([torch.arange(1, n), torch.zeros(n - 1, dtype=torch.int64)])
    keep = (torch.arange(2 * (n - 1)) % 7) != 3          # some rows lose their only edge
    gg = graph.graph_from_edges(src[keep], dst[keep], n)
    pp, p2n = _lib.build_part(4, gg.row_pointers)
    X = torch.randn(n, 64, generator=torch.Generator().manual_seed(4))
    for slack, dynamic in ((1, 1), (2, 0), (1000, 1)):
        with sweep_forced(6, slack=slack, dynamic=dynamic, gcn_prescale=1) as s:
            check_modes(gg, X, pp, p2n, 4, what=f"sweep hub slack={slack}", sum_scale=True)
            assert s.launches() == 3


def test_sweep_non_canonical_partition_is_still_correct():
    g, X, pp, p2n = make_case(1500, 90000, 64, 8, seed=11, kind="powerlaw")
    P = p2n.numel()
    perm = torch.randperm(P, generator=torch.Generator().manual_seed(5))
    # shuffled groups: each keeps its edge range, the order (and therefore part2Node) is no longer monotone
    beg, end = pp[:-1][perm], pp[1:][perm]
    lens = (end - beg).to(torch.int64)
    ci_new = torch.cat([g.column_index[int(b):int(e)] for b, e in zip(beg.tolist(), end.tolist())])
    pp_new = torch.zeros(P + 1, dtype=torch.int32)
    pp_new[1:] = torch.cumsum(lens, 0).to(torch.int32)
    p2n_new = p2n[perm].contiguous()
    Xd, cid, ppd, p2nd = dev(X, ci_new.contiguous(), pp_new, p2n_new)
    with sweep_forced(4) as s:
        y = _lib.sag(Xd, None, cid, None, ppd, p2nd, 8, 32, 4).cpu().numpy()
        assert s.launches() == 1
    assert_close_f64(y, oracle.csr_f64(0, X.numpy(), g.row_pointers.numpy(), g.column_index.numpy()),
                     what="sweep on shuffled groups")


def test_sweep_rectangular_and_accumulate():
    """Destination shard: out rows != source rows, and a second call that adds into the first one's result."""
    n_out, n_in, dim, ps = 1200, 9000, 64, 16
    rp, ci = graph.powerlaw_shard(n_out, n_in, 150000, 4000, seed=3)
    pp, p2n = _lib.build_part(ps, rp)
    X = torch.randn(n_in, dim, generator=torch.Generator().manual_seed(8))
    Xd, cid, ppd, p2nd = dev(X, ci, pp, p2n)
    ref = oracle.csr_f64(0, X.numpy(), rp.numpy(), ci.numpy())
    with sweep_forced(8) as s:
        y = _lib.agg_rect(0, Xd, cid, ppd, p2nd, n_out, ps)
        y2 = _lib.agg_rect(0, Xd, cid, ppd, p2nd, n_out, ps, out=y.clone(), accumulate=True)
        assert s.launches() == 2
    assert_close_f64(y.cpu().numpy(), ref, what="sweep rect")
    assert_close_f64(y2.cpu().numpy(), 2 * ref, what="sweep rect accumulate")


def test_sweep_in_a_captured_graph_after_prepare():
    """gnna_prepare_graph makes the plan up front: the call inside a stream capture takes the sliced (here: sweep)
    schedule, and replays give the same result."""
    g, X, pp, p2n = make_case(6000, 700000, 64, 32, seed=21, kind="powerlaw")
    Xd, rp, ci, deg, ppd, p2nd = dev(X, g.row_pointers, g.column_index, g.degrees, pp, p2n)
    out = torch.empty_like(Xd)
    _lib.reset_tuning()
    _lib.set_tuning(sweep=1, column_phases=8, deterministic=0)
    try:
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            _lib.prepare_graph(ci, ppd, p2nd, g.num_nodes, g.num_nodes, 32, [64])
            _lib.sag(Xd, rp, ci, deg, ppd, p2nd, 32, 32, 4, out=out)      # warm-up (scratch)
            side.synchronize()
            before = _lib.runtime_counters()
            graph_ = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph_, stream=side):
                _lib.sag(Xd, rp, ci, deg, ppd, p2nd, 32, 32, 4, out=out)
                assert _lib.last_num_phases() == 8
        after = _lib.runtime_counters()
        assert after["sweep_launches"] == before["sweep_launches"] + 1
        for k in ("plan_builds", "launch_syncs", "launch_frees", "launch_mallocs"):
            assert after[k] == before[k], k
        out.fill_(float("nan"))
        graph_.replay()
        torch.cuda.synchronize()
        assert_close_f64(out.cpu().numpy(), oracle.csr_f64(0, X.numpy(), g.row_pointers.numpy(), g.column_index.numpy()),
                         what="sweep replay")
    finally:
        _lib.reset_tuning()
        _lib.release_graph(ci)


def test_two_captured_sweep_graphs_replayed_side_by_side():
    """the sweep kernel's step counters and lists live in a per-call block of device scratch; captured calls used to
    take theirs from a ring by sequence number, so two graphs replayed concurrently on different streams could land in one block
    (a hang or wrong rows).  Since round 6 every captured call keeps a block of its own for good: two graphs, each one sweep
    aggregation on its own graph, replayed together forty times on two streams, both results right every time."""
    cases = []
    _lib.reset_tuning()
    _lib.set_tuning(sweep=1, column_phases=8, deterministic=0)
    try:
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        for k, s in enumerate(streams):
            g, X, pp, p2n = make_case(6000, 700000, 64, 32, seed=31 + k, kind="powerlaw")
            Xd, rp, ci, deg, ppd, p2nd = dev(X, g.row_pointers, g.column_index, g.degrees, pp, p2n)
            out = torch.empty_like(Xd)
            with torch.cuda.stream(s):
                _lib.prepare_graph(ci, ppd, p2nd, g.num_nodes, g.num_nodes, 32, [64])
                _lib.sag(Xd, rp, ci, deg, ppd, p2nd, 32, 32, 4, out=out)      # warm-up (scratch)
                s.synchronize()
                before = _lib.runtime_counters()["sweep_launches"]
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=s):
                    _lib.sag(Xd, rp, ci, deg, ppd, p2nd, 32, 32, 4, out=out)
                assert _lib.runtime_counters()["sweep_launches"] == before + 1
            ref = oracle.csr_f64(0, X.numpy(), g.row_pointers.numpy(), g.column_index.numpy())
            cases.append((gr, out, ref, ci, (Xd, rp, deg, ppd, p2nd)))
        torch.cuda.synchronize()
        for rep in range(40):
            for (gr, out, _, _, _), s in zip(cases, streams):
                out.fill_(float("nan"))
            torch.cuda.synchronize()
            for (gr, _, _, _, _), s in zip(cases, streams):
                with torch.cuda.stream(s):
                    gr.replay()
            torch.cuda.synchronize()
            if rep % 8 == 0 or rep == 39:
                for k, (_, out, ref, _, _) in enumerate(cases):
                    assert_close_f64(out.cpu().numpy(), ref, what=f"graph {k}, concurrent replay {rep}")
    finally:
        _lib.reset_tuning()
        for c in cases:
            _lib.release_graph(c[3])
