"""Packed column ids of a prepared graph (gnna_prepare_graph + gnna_tuning.pack_ids): the sliced schedule reads a
plan-owned copy of the ids in the order it consumes them.  Same parity bars as test_parity_gpu.py (X = ones exact --
reference unitest.py:27,54-63 -- and random inputs within 1e-4 * scale of the fp64 CSR formula), every case checks
through gnna_runtime_counters that the packed path really ran."""
import numpy as np
import pytest
import torch

import oracle
from gnnadvisor_osdi21_amd import _lib, graph
from util import assert_close_f64, dev, make_case

pytestmark = pytest.mark.gpu


def counters():
    c = _lib.runtime_counters()
    return c["pack_builds"], c["packed_launches"]


def check_all_modes(g, X, pp, p2n, ps, Xd, rp, ci, deg, ppd, p2nd, what):
    Xn, cin, rpn, degn = X.numpy(), g.column_index.numpy(), g.row_pointers.numpy(), g.degrees.numpy()
    ys = _lib.sag(Xd, rp, ci, deg, ppd, p2nd, ps, 32, 4)
    yg = _lib.agg_gcn(Xd, rp, ci, deg, ppd, p2nd, ps, 32, 4)
    yi = _lib.agg_gin(Xd, rp, ci, 0.5, ppd, p2nd, ps, 32, 4)
    assert_close_f64(ys.cpu().numpy(), oracle.csr_f64(0, Xn, rpn, cin), what=what + " sag")
    assert_close_f64(yg.cpu().numpy(), oracle.csr_f64(1, Xn, rpn, cin, degn), what=what + " gcn",
                     scale=oracle.csr_f64(1, np.abs(Xn), rpn, cin, degn))
    assert_close_f64(yi.cpu().numpy(), oracle.csr_f64(2, Xn, rpn, cin, None, 0.5), what=what + " gin")
    assert_close_f64(ys.cpu().numpy(), oracle.sag(Xn, cin, pp.numpy(), p2n.numpy()), what=what + " sag vs oracle")


@pytest.mark.parametrize("dim", [4, 7, 16, 41, 64, 100, 128, 300])
@pytest.mark.parametrize("phases,ps,gpc", [(2, 16, 16), (5, 3, 16), (8, 64, 16), (32, 32, 16), (8, 100, 4), (3, 1, 5)])
def test_packed_ids_match_the_oracle(dim, phases, ps, gpc):
    g, X, pp, p2n = make_case(3000, 200000, dim, ps, seed=dim * 3 + phases, kind="powerlaw")
    Xd, rp, ci, deg, ppd, p2nd = dev(X, g.row_pointers, g.column_index, g.degrees, pp, p2n)
    _lib.reset_tuning()
    _lib.set_tuning(column_phases=phases, groups_per_chunk=gpc, deterministic=0, sweep=0, wide_blocks=2)
    try:
        for prescale in (1, 2):        # pre-scaled and per-edge GCN forms
            _lib.set_tuning(gcn_prescale=prescale)
            got = _lib.prepare_graph(ci, ppd, p2nd, g.num_nodes, g.num_nodes, ps, [dim])
            assert got[dim] == phases
            b0, l0 = counters()
            check_all_modes(g, X, pp, p2n, ps, Xd, rp, ci, deg, ppd, p2nd, f"packed dim={dim} phases={phases} ps={ps}")
            b1, l1 = counters()
            assert l1 - l0 == 3, "the packed path did not run"
            assert b1 == b0, "a copy was built inside an aggregation although prepare had the width"
    finally:
        _lib.reset_tuning()
        _lib.release_graph(ci)


def test_packed_ones_is_exact_and_off_switch():
    g, X, pp, p2n = make_case(5000, 600000, 64, 32, seed=3, kind="powerlaw", x="ones")
    Xd, rp, ci, deg, ppd, p2nd = dev(X, g.row_pointers, g.column_index, g.degrees, pp, p2n)
    want = (g.row_pointers[1:] - g.row_pointers[:-1]).to(torch.float32)[:, None].expand(-1, 64)
    _lib.reset_tuning()
    try:
        for phases in (2, 8, 16, 32):
            _lib.set_tuning(column_phases=phases)
            _lib.prepare_graph(ci, ppd, p2nd, g.num_nodes, g.num_nodes, 32, [64])
            b0, l0 = counters()
            y = _lib.sag(Xd, rp, ci, deg, ppd, p2nd, 32, 32, 4).cpu()
            assert counters()[1] == l0 + 1
            assert torch.equal(y, want), phases
        _lib.set_tuning(pack_ids=2)
        l0 = counters()[1]
        y = _lib.sag(Xd, rp, ci, deg, ppd, p2nd, 32, 32, 4).cpu()
        assert counters()[1] == l0 and torch.equal(y, want)
    finally:
        _lib.reset_tuning()
        _lib.release_graph(ci)


def test_more_phase_counts_than_copies_and_release():
    """Four copies per graph: a fifth phase count in turn with the others gets none (a copy in recent use is not replaced:
    no build storm) and reads column_index; once the others have gone unused for a while it takes the oldest one's place;
    after the release the ids are read from column_index again."""
    if _lib.get_tuning()["pack_ids"] != 0:
        pytest.skip("GNNA_TUNE changes the packing policy: the copy bookkeeping is not under test")
    g, X, pp, p2n = make_case(4000, 300000, 64, 16, seed=9, kind="powerlaw")
    Xd, rp, ci, deg, ppd, p2nd = dev(X, g.row_pointers, g.column_index, g.degrees, pp, p2n)
    ref = oracle.csr_f64(0, X.numpy(), g.row_pointers.numpy(), g.column_index.numpy())
    _lib.reset_tuning()
    try:
        _lib.prepare_graph(ci, ppd, p2nd, g.num_nodes, g.num_nodes, 16, [])
        b0, l0 = counters()
        for rep in range(2):
            for phases in (2, 4, 8, 16, 32):
                _lib.set_tuning(column_phases=phases)
                y = _lib.sag(Xd, rp, ci, deg, ppd, p2nd, 16, 32, 4)
                assert_close_f64(y.cpu().numpy(), ref, what=f"phases {phases}")
        b1, l1 = counters()
        assert b1 - b0 == 4 and l1 - l0 == 8            # 32 phases ran without a copy, twice
        for _ in range(20):                              # the other copies age ...
            y = _lib.sag(Xd, rp, ci, deg, ppd, p2nd, 16, 32, 4)
        b2, l2 = counters()
        assert b2 - b1 == 1 and l2 - l1 >= 8            # ... and 32 phases gets the oldest one's place
        assert_close_f64(y.cpu().numpy(), ref, what="phases 32, packed")
        _lib.release_graph(ci)
        l0 = counters()[1]
        y = _lib.sag(Xd, rp, ci, deg, ppd, p2nd, 16, 32, 4)
        assert counters()[1] == l0
        assert_close_f64(y.cpu().numpy(), ref, what="after release")
    finally:
        _lib.reset_tuning()
        _lib.release_graph(ci)


def test_packed_non_canonical_rectangular_accumulate_and_deterministic():
    # shuffled groups (part2Node not monotone), a destination shard with its own source range, accumulate, and the
    # ordered schedule: all read the same packed copies
    g, X, pp, p2n = make_case(1500, 90000, 64, 8, seed=11, kind="powerlaw")
    P = p2n.numel()
    perm = torch.randperm(P, generator=torch.Generator().manual_seed(5))
    beg, end = pp[:-1][perm], pp[1:][perm]
    ci_new = torch.cat([g.column_index[int(b):int(e)] for b, e in zip(beg.tolist(), end.tolist())]).contiguous()
    pp_new = torch.zeros(P + 1, dtype=torch.int32)
    pp_new[1:] = torch.cumsum((end - beg).to(torch.int64), 0).to(torch.int32)
    p2n_new = p2n[perm].contiguous()
    Xd, cid, ppd, p2nd = dev(X, ci_new, pp_new, p2n_new)
    ref = oracle.csr_f64(0, X.numpy(), g.row_pointers.numpy(), g.column_index.numpy())
    _lib.reset_tuning()
    _lib.set_tuning(column_phases=4)
    try:
        _lib.prepare_graph(cid, ppd, p2nd, g.num_nodes, g.num_nodes, 8, [64])
        l0 = counters()[1]
        y = _lib.sag(Xd, None, cid, None, ppd, p2nd, 8, 32, 4)
        assert counters()[1] == l0 + 1
        assert_close_f64(y.cpu().numpy(), ref, what="packed, shuffled groups")
    finally:
        _lib.release_graph(cid)
    n_out, n_in = 1200, 9000
    rp, ci = graph.powerlaw_shard(n_out, n_in, 150000, 4000, seed=3)
    pp2, p2n2 = _lib.build_part(16, rp)
    X2 = torch.randn(n_in, 64, generator=torch.Generator().manual_seed(8))
    X2d, ci2d, pp2d, p2n2d = dev(X2, ci, pp2, p2n2)
    ref2 = oracle.csr_f64(0, X2.numpy(), rp.numpy(), ci.numpy())
    try:
        _lib.set_tuning(column_phases=8)
        _lib.prepare_graph(ci2d, pp2d, p2n2d, n_in, n_out, 16, [64])
        l0 = counters()[1]
        y = _lib.agg_rect(0, X2d, ci2d, pp2d, p2n2d, n_out, 16)
        y2 = _lib.agg_rect(0, X2d, ci2d, pp2d, p2n2d, n_out, 16, out=y.clone(), accumulate=True)
        _lib.set_tuning(deterministic=1)
        y3 = _lib.agg_rect(0, X2d, ci2d, pp2d, p2n2d, n_out, 16)
        y4 = _lib.agg_rect(0, X2d, ci2d, pp2d, p2n2d, n_out, 16)
        assert counters()[1] == l0 + 4
        assert_close_f64(y.cpu().numpy(), ref2, what="packed rect")
        assert_close_f64(y2.cpu().numpy(), 2 * ref2, what="packed rect accumulate")
        assert_close_f64(y3.cpu().numpy(), ref2, what="packed rect deterministic")
        assert torch.equal(y3, y4)
    finally:
        _lib.reset_tuning()
        _lib.release_graph(ci2d)


def test_pack_ids_1_packs_unprepared_graphs_too_and_evicted_plans_drop_their_copies():
    """pack_ids = 1: the caller's promise covers every graph -- automatic plans get copies at first use; 40 graphs through
    the 32 automatic plans: an evicted plan's copies go with it and every result stays right."""
    _lib.reset_tuning()
    _lib.release_graph(None)
    _lib.set_tuning(column_phases=4, pack_ids=1)
    keep = []
    try:
        for i in range(40):
            g, X, pp, p2n = make_case(1200, 60000, 32, 8, seed=100 + i, kind="powerlaw")
            Xd, rp, ci, deg, ppd, p2nd = dev(X, g.row_pointers, g.column_index, g.degrees, pp, p2n)
            keep.append((g, X, Xd, rp, ci, deg, ppd, p2nd))
        for rep in range(2):
            b0, l0 = counters()
            for g, X, Xd, rp, ci, deg, ppd, p2nd in keep:
                y = _lib.sag(Xd, rp, ci, deg, ppd, p2nd, 8, 32, 4)
                assert_close_f64(y.cpu().numpy(), oracle.csr_f64(0, X.numpy(), g.row_pointers.numpy(), g.column_index.numpy()),
                                 what="pack_ids=1, unprepared")
            b1, l1 = counters()
            assert l1 - l0 == 40 and b1 - b0 == 40       # 40 graphs cycle through 32 plans: every call rebuilds
    finally:
        _lib.reset_tuning()
        _lib.release_graph(None)


@pytest.mark.parametrize("sweep", [0, 1])
def test_a_rewritten_column_index_is_noticed_and_the_copy_bypassed(sweep):
    """The packed copy exists under the caller's promise not to change the graph; a broken promise is still caught when it
    is wholesale: every call compares 1024 samples of column_index with what the copy was made from and reads
    column_index itself when they differ (the slice counts may be stale then -- that costs locality, not correctness)."""
    g, X, pp, p2n = make_case(4000, 300000, 64, 16, seed=31, kind="powerlaw")
    Xd, rp, ci, deg, ppd, p2nd = dev(X, g.row_pointers, g.column_index, g.degrees, pp, p2n)
    _lib.reset_tuning()
    _lib.set_tuning(column_phases=8, sweep=sweep)
    try:
        _lib.prepare_graph(ci, ppd, p2nd, g.num_nodes, g.num_nodes, 16, [64])
        l0 = counters()[1]
        y = _lib.sag(Xd, rp, ci, deg, ppd, p2nd, 16, 32, 4)
        assert counters()[1] == l0 + 1
        assert_close_f64(y.cpu().numpy(), oracle.csr_f64(0, X.numpy(), g.row_pointers.numpy(), g.column_index.numpy()),
                         what="before the rewrite")
        # same row lengths, other neighbours, written into the SAME device tensor
        new_ci = (g.column_index.to(torch.int64) * 7 + 13).remainder(g.num_nodes).to(torch.int32)
        ci.copy_(new_ci.to(ci.device))
        for _ in range(2):
            y = _lib.sag(Xd, rp, ci, deg, ppd, p2nd, 16, 32, 4)
            assert_close_f64(y.cpu().numpy(), oracle.csr_f64(0, X.numpy(), g.row_pointers.numpy(), new_ci.numpy()),
                             what="after the rewrite")
        ci.copy_(g.column_index.to(ci.device))          # and back: the copy is valid again
        y = _lib.sag(Xd, rp, ci, deg, ppd, p2nd, 16, 32, 4)
        assert_close_f64(y.cpu().numpy(), oracle.csr_f64(0, X.numpy(), g.row_pointers.numpy(), g.column_index.numpy()),
                         what="restored")
    finally:
        _lib.reset_tuning()
        _lib.release_graph(ci)


def test_packed_in_a_captured_graph():
    g, X, pp, p2n = make_case(6000, 700000, 64, 32, seed=21, kind="powerlaw")
    Xd, rp, ci, deg, ppd, p2nd = dev(X, g.row_pointers, g.column_index, g.degrees, pp, p2n)
    out = torch.empty_like(Xd)
    _lib.reset_tuning()
    _lib.set_tuning(column_phases=8)
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
        after = _lib.runtime_counters()
        assert after["packed_launches"] == before["packed_launches"] + 1
        for k in ("plan_builds", "pack_builds", "launch_syncs", "launch_frees", "launch_mallocs"):
            assert after[k] == before[k], k
        out.fill_(float("nan"))
        graph_.replay()
        torch.cuda.synchronize()
        assert_close_f64(out.cpu().numpy(), oracle.csr_f64(0, X.numpy(), g.row_pointers.numpy(), g.column_index.numpy()),
                         what="packed replay")
    finally:
        _lib.reset_tuning()
        _lib.release_graph(ci)


@pytest.mark.parametrize("dim,partSize,phases", [(64, 64, 0), (64, 128, 16), (32, 32, 8), (41, 64, 4), (16, 16, 8), (100, 64, 4), (128, 32, 0)])
def test_sddmm_on_a_prepared_graph_reads_the_packed_ids(dim, partSize, phases):
    """Round 6: SDDMM reads the packed copy of a prepared graph's ids too -- the kernel carries every piece's place in the copy
    AND its original position, because edge_out is indexed like column_index.  Against the dense formula, every edge; then three
    ids are rewritten behind the library's back (`.data`): with the full hash at every call the very next result is the edited
    graph's."""
    import numpy as np
    import oracle
    from util import assert_close_f64
    g = graph.powerlaw_graph(30000, 3000000, 3000, seed=dim + partSize, device="cuda")
    n = g.num_nodes
    ci = g.column_index.clone()
    pp, p2n = [t.cuda() for t in _lib.build_part(partSize, g.row_pointers.cpu())]
    gen = torch.Generator(device="cuda").manual_seed(dim)
    A = torch.randn(n, dim, device="cuda", generator=gen)
    B = torch.randn(n, dim, device="cuda", generator=gen)
    rp_h = g.row_pointers.cpu().numpy()
    rows = np.repeat(np.arange(n), np.diff(rp_h))

    rows_d = torch.from_numpy(rows).cuda()

    def check(what):
        # the dense formula in fp64 on the device for all 3 M edges (the numpy oracle holds it on a sample: its [E, D] fp64
        # temporaries are GBs at this size)
        cl = ci.long()
        ref = torch.zeros(cl.numel(), dtype=torch.float64, device="cuda")
        scale = torch.zeros_like(ref)
        for c0 in range(0, cl.numel(), 1 << 20):
            a, b = A[rows_d[c0:c0 + (1 << 20)]].double(), B[cl[c0:c0 + (1 << 20)]].double()
            ref[c0:c0 + (1 << 20)] = (a * b).sum(1)
            scale[c0:c0 + (1 << 20)] = (a.abs() * b.abs()).sum(1)
        lo = cl.numel() // 3
        hi = lo + 20000
        rp_s = np.clip(rp_h.astype(np.int64), lo, hi) - lo                   # row pointers of the edges [lo, hi) alone
        ora = oracle.np_sddmm(A.cpu().numpy(), B.cpu().numpy(), rp_s, ci[lo:hi].cpu().numpy())
        assert np.allclose(ora, ref[lo:hi].cpu().numpy(), rtol=1e-9, atol=1e-9)
        out = _lib.sddmm(A, B, ci, pp, p2n, partSize)
        assert_close_f64(out.cpu().numpy(), ref.cpu().numpy(), what=f"{what}: sddmm dim={dim} ps={partSize} phases={phases}",
                         scale=scale.cpu().numpy(), rtol=1e-5)
    try:
        _lib.reset_tuning()
        _lib.set_tuning(column_phases=phases if phases else -1, ids_check_every=1, nonlocal_ids=1, avg_degree=100)
        check("unprepared")
        _lib.prepare_graph(ci, pp, p2n, n, n, partSize, [dim])
        packed0 = _lib.runtime_counters()["packed_launches"]
        check("prepared")
        check("prepared, again")
        used_packed = _lib.runtime_counters()["packed_launches"] > packed0
        if phases >= 2:
            assert used_packed, "a sliced SDDMM on a prepared graph must read the packed copy"
        for r in (3, 1717, n - 5):
            b = int(g.row_pointers[r])
            if int(g.row_pointers[r + 1]) > b:
                ci.data[b] = (int(ci[b]) + 11) % n
        check("three ids edited behind the library's back")
        check("... and the call after")
    finally:
        _lib.release_graph(ci)
        _lib.reset_tuning()
