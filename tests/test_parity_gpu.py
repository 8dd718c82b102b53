"""GPU parity: the HIP path (through the C ABI of libgnna.so) against the CPU oracle.

Tolerances: X = ones (the reference's own known-answer test, unitest.py:27,54-63) must be
bit-exact; random fp32 inputs within 1e-4 * max(1, |ref|) of the fp64 CSR formula
(north_star: "within 1e-4 fp32"), and within the same bound of the fp32 oracle.
"""
import json
import os

import numpy as np
import pytest
import torch

import oracle
from gnnadvisor_osdi21_amd import _lib, graph
from util import assert_close_f64, dev, make_case

pytestmark = pytest.mark.gpu

DIMS = [1, 2, 3, 4, 6, 7, 8, 16, 32, 41, 64, 100, 128, 256, 300, 602, 1433, 3703]


def run_all_modes(g, X, pp, p2n, partSize, eps=0.5):
    Xd, rp, ci, deg, ppd, p2nd = dev(X, g.row_pointers, g.column_index, g.degrees, pp, p2n)
    ys = _lib.sag(Xd, rp, ci, deg, ppd, p2nd, partSize, 32, 4)
    yg = _lib.agg_gcn(Xd, rp, ci, deg, ppd, p2nd, partSize, 32, 4)
    yi = _lib.agg_gin(Xd, rp, ci, eps, ppd, p2nd, partSize, 32, 4)
    torch.cuda.synchronize()
    return ys.cpu().numpy(), yg.cpu().numpy(), yi.cpu().numpy()


def check_all_modes(g, X, pp, p2n, partSize, eps=0.5, what=""):
    ys, yg, yi = run_all_modes(g, X, pp, p2n, partSize, eps)
    Xn, ci, ppn, p2nn = X.numpy(), g.column_index.numpy(), pp.numpy(), p2n.numpy()
    rp, deg = g.row_pointers.numpy(), g.degrees.numpy()
    assert_close_f64(ys, oracle.csr_f64(0, Xn, rp, ci), what=what + " sag vs fp64")
    gscale = oracle.csr_f64(1, np.abs(Xn), rp, ci, deg)
    assert_close_f64(yg, oracle.csr_f64(1, Xn, rp, ci, deg), what=what + " gcn vs fp64", scale=gscale)
    assert_close_f64(yi, oracle.csr_f64(2, Xn, rp, ci, None, eps), what=what + " gin vs fp64")
    assert_close_f64(ys, oracle.sag(Xn, ci, ppn, p2nn), what=what + " sag vs oracle")
    assert_close_f64(yg, oracle.gcn_aggregate(Xn, ci, deg, ppn, p2nn), what=what + " gcn vs oracle", scale=gscale)
    assert_close_f64(yi, oracle.gin_aggregate(Xn, ci, eps, ppn, p2nn), what=what + " gin vs oracle")


@pytest.mark.parametrize("dim", DIMS)
def test_dims(dim):
    g, X, pp, p2n = make_case(300, 6000, dim, 32, seed=dim)
    check_all_modes(g, X, pp, p2n, 32, what=f"dim={dim}")


@pytest.mark.parametrize("partSize", [1, 2, 3, 7, 16, 32, 64, 100, 491])
def test_part_sizes(partSize):
    g, X, pp, p2n = make_case(500, 40000, 64, partSize, seed=partSize, kind="powerlaw")
    check_all_modes(g, X, pp, p2n, partSize, what=f"ps={partSize}")


@pytest.mark.parametrize("G,U,bpc,xcd", [(1, 4, 0, 0), (3, 8, 0, 1), (16, 16, 0, 1), (63, 8, 0, 0),
                                         (16, 8, 1, 1), (8, 4, 2, 0), (32, 8, 8, 1)])
def test_scheduling_knobs_do_not_change_results(G, U, bpc, xcd):
    g, X, pp, p2n = make_case(2000, 120000, 64, 8, seed=7, kind="powerlaw")
    try:
        _lib.set_tuning(G, U, bpc, xcd, 0)
        check_all_modes(g, X, pp, p2n, 8, what=f"G={G} U={U} bpc={bpc} xcd={xcd}")
    finally:
        _lib.reset_tuning()


def test_kat_ones_exact_golden(golden_dir):
    """The reference's own test: X = ones => every column equals the row's nnz, exactly."""
    cases = json.load(open(os.path.join(golden_dir, "kat_ones.json")))["cases"]
    for c in cases:
        n, dim, ps = c["num_nodes"], c["dim"], c["partSize"]
        rp = torch.tensor(c["row_pointers"], dtype=torch.int32)
        ci = torch.tensor(c["column_index"], dtype=torch.int32)
        pp, p2n = _lib.build_part(ps, rp)
        X = torch.ones(n, dim)
        deg = graph.degrees_from_rowptr(rp)
        Xd, rpd, cid, degd, ppd, p2nd = dev(X, rp, ci, deg, pp, p2n)
        y = _lib.sag(Xd, rpd, cid, degd, ppd, p2nd, ps, 32, 4).cpu().numpy()
        want = np.repeat(np.asarray(c["expected_row_value"], dtype=np.float32)[:, None], dim, 1)
        assert np.array_equal(y, want), c["name"]


def test_appendix_a_worked_example():
    rp = torch.tensor([0, 3, 4, 6, 7], dtype=torch.int32)
    ci = torch.tensor([1, 2, 3, 0, 0, 3, 2], dtype=torch.int32)
    pp, p2n = _lib.build_part(2, rp)
    X = torch.tensor([[0., 1.], [2., 3.], [4., 5.], [6., 7.]])
    deg = graph.degrees_from_rowptr(rp)
    Xd, rpd, cid, degd, ppd, p2nd = dev(X, rp, ci, deg, pp, p2n)
    ys = _lib.sag(Xd, rpd, cid, degd, ppd, p2nd, 2, 32, 4).cpu().numpy()
    yg = _lib.agg_gcn(Xd, rpd, cid, degd, ppd, p2nd, 2, 32, 4).cpu().numpy()
    yi = _lib.agg_gin(Xd, rpd, cid, 0.5, ppd, p2nd, 2, 32, 4).cpu().numpy()
    assert np.array_equal(ys, np.array([[12, 15], [0, 1], [6, 8], [4, 5]], dtype=np.float32))
    np.testing.assert_allclose(yg, [[23.65437, 29.56796], [0, 1.73205], [8.48528, 12.34898],
                                    [5.65685, 7.07107]], rtol=1e-5)
    assert np.array_equal(yi, np.array([[6, 7.5], [0, 0.5], [3, 4], [2, 2.5]], dtype=np.float32))


def test_edge_cases_empty_and_ragged():
    # no edges at all: output must be all zeros (fresh zeros_like in the reference)
    g, X, pp, p2n = make_case(50, 0, 16, 4, seed=1)
    ys, yg, yi = run_all_modes(g, X, pp, p2n, 4)
    assert not ys.any() and not yg.any() and not yi.any()
    # many zero-degree rows, including the last one (the reference's missing-sentinel case)
    rp = torch.tensor([0, 3, 3, 8, 9, 9], dtype=torch.int32)
    ci = torch.tensor([1, 2, 4, 0, 1, 2, 3, 4, 0], dtype=torch.int32)
    for ps in (1, 2, 3, 32):
        pp, p2n = _lib.build_part(ps, rp)
        gg = graph.CSRGraph(5, rp, ci, graph.degrees_from_rowptr(rp), 9, 9 / 5, 0.0)
        X = torch.randn(5, 8, generator=torch.Generator().manual_seed(3))
        check_all_modes(gg, X, pp, p2n, ps, what=f"ragged ps={ps}")
    # one hub row holding almost every edge, split over many chunks (atomic flush path)
    n = 3000
    src = torch.cat([torch.zeros(n - 1, dtype=torch.int64), torch.arange(1, n)])
    dst = torch.cat([torch.arange(1, n), torch.zeros(n - 1, dtype=torch.int64)])
    gg = graph.graph_from_edges(src, dst, n)
    pp, p2n = _lib.build_part(4, gg.row_pointers)
    X = torch.randn(n, 64, generator=torch.Generator().manual_seed(4))
    check_all_modes(gg, X, pp, p2n, 4, what="hub")
    # zero nodes
    e = torch.zeros(0, dtype=torch.int64)
    g0 = graph.graph_from_edges(e, e, 0)
    pp, p2n = _lib.build_part(4, g0.row_pointers)
    y = _lib.sag(torch.zeros(0, 8).cuda(), g0.row_pointers.cuda(), g0.column_index.cuda(),
                 g0.degrees.cuda(), pp.cuda(), p2n.cuda(), 4, 32, 4)
    assert y.shape == (0, 8)


def test_non_canonical_partition_is_still_correct():
    """Groups shuffled out of row order (never produced by build_part): the validation pass
    must route every flush through atomics."""
    g, X, pp, p2n = make_case(400, 20000, 64, 5, seed=11)
    P = p2n.numel()
    perm = torch.randperm(P, generator=torch.Generator().manual_seed(5))
    # a partition is (begin, end, row) per group; shuffling needs explicit ends, so build a
    # permuted CSR instead: re-order rows' groups by permuting whole groups of equal size
    beg, end = pp[:-1].clone(), pp[1:].clone()
    # keep the contiguous-offset contract: rebuild column_index in the permuted group order
    new_ci, new_pp, new_p2n = [], [0], []
    for k in perm.tolist():
        seg = g.column_index[beg[k]:end[k]]
        new_ci.append(seg)
        new_pp.append(new_pp[-1] + seg.numel())
        new_p2n.append(int(p2n[k]))
    ci2 = torch.cat(new_ci).to(torch.int32)
    pp2 = torch.tensor(new_pp, dtype=torch.int32)
    p2n2 = torch.tensor(new_p2n, dtype=torch.int32)
    Xd, rp, cid, deg, ppd, p2nd = dev(X, g.row_pointers, ci2, g.degrees, pp2, p2n2)
    ys = _lib.sag(Xd, rp, cid, deg, ppd, p2nd, 5, 32, 4).cpu().numpy()
    yg = _lib.agg_gcn(Xd, rp, cid, deg, ppd, p2nd, 5, 32, 4).cpu().numpy()
    Xn = X.numpy()
    assert_close_f64(ys, oracle.csr_f64(0, Xn, g.row_pointers.numpy(), g.column_index.numpy()), what="shuffled sag")
    assert_close_f64(yg, oracle.csr_f64(1, Xn, g.row_pointers.numpy(), g.column_index.numpy(), g.degrees.numpy()),
                     what="shuffled gcn",
                     scale=oracle.csr_f64(1, np.abs(Xn), g.row_pointers.numpy(), g.column_index.numpy(), g.degrees.numpy()))
    assert_close_f64(ys, oracle.sag(Xn, ci2.numpy(), pp2.numpy(), p2n2.numpy()), what="shuffled sag vs oracle")


def test_dword_aligned_views():
    g, X, pp, p2n = make_case(200, 5000, 64, 16, seed=21)
    Xd, rp, ci, deg, ppd, p2nd = dev(X, g.row_pointers, g.column_index, g.degrees, pp, p2n)
    buf = torch.zeros(X.numel() + 1, device="cuda")
    Xu = buf[1:].view(X.shape)          # 4-byte aligned only
    Xu.copy_(Xd)
    assert Xu.data_ptr() % 16 != 0
    y = _lib.sag(Xu, rp, ci, deg, ppd, p2nd, 16, 32, 4).cpu().numpy()
    assert_close_f64(y, oracle.csr_f64(0, X.numpy(), g.row_pointers.numpy(), g.column_index.numpy()), what="unaligned")


def test_output_is_fully_overwritten_and_repeatable():
    g, X, pp, p2n = make_case(1000, 50000, 64, 32, seed=31, kind="powerlaw", x="ones")
    Xd, rp, ci, deg, ppd, p2nd = dev(X, g.row_pointers, g.column_index, g.degrees, pp, p2n)
    out = torch.full_like(Xd, float("nan"))
    for _ in range(3):
        _lib.sag(Xd, rp, ci, deg, ppd, p2nd, 32, 32, 4, out=out)
    want = (g.row_pointers[1:] - g.row_pointers[:-1]).float()[:, None].expand(-1, 64)
    assert torch.equal(out.cpu(), want)


@pytest.mark.parametrize("G", [1, 3, 16, 64])
@pytest.mark.parametrize("dim", [4, 6, 64, 100, 300])
def test_sparse_zero_fill_clears_exactly_the_rows_nobody_stores(G, dim):
    """zero_fill = 1 (automatic from 32 MiB of output): a single pass of the streaming kernel stores the rows it
    owns, the prologue clears only rows without edges, rows of empty groups and rows that continue from the previous
    work item.  Every output starts as NaN: a row neither cleared nor stored -- or cleared after being stored --
    shows."""
    rng = np.random.default_rng(100 * G + dim)
    n = 700
    deg = rng.choice([0, 0, 1, 2, 5, 40], size=n)
    deg[:3] = 0; deg[-4:] = 0; deg[300:340] = 0                     # runs of rows without edges: first, last, middle
    deg[50] = 600                                                   # a hub that spans many work items
    rp = np.zeros(n + 1, dtype=np.int32); rp[1:] = np.cumsum(deg)
    ci = np.concatenate([np.sort(rng.choice(n, size=d, replace=False)) for d in deg]).astype(np.int32)
    X = torch.randn(n, dim, generator=torch.Generator().manual_seed(dim))
    Xn = X.numpy()
    ref = oracle.csr_f64(0, Xn, rp, ci)
    rpt, cit = torch.from_numpy(rp), torch.from_numpy(ci)
    degrees = graph.degrees_from_rowptr(rpt)
    try:
        _lib.set_tuning(groups_per_chunk=G, zero_fill=1)
        for ps in (1, 4, 32):
            pp, p2n = _lib.build_part(ps, rpt)
            Xd, rpd, cid, degd, ppd, p2nd = dev(X, rpt, cit, degrees, pp, p2n)
            out = torch.full((n, dim), float("nan"), device="cuda")
            _lib.sag(Xd, rpd, cid, degd, ppd, p2nd, ps, 32, 4, out=out)
            assert_close_f64(out.cpu().numpy(), ref, what=f"sparse zero-fill ps={ps}")
            out.fill_(float("nan"))
            _lib.agg_gcn(Xd, rpd, cid, degd, ppd, p2nd, ps, 32, 4, out=out)
            assert_close_f64(out.cpu().numpy(), oracle.csr_f64(1, Xn, rp, ci, degrees.numpy()), what=f"sparse zero-fill gcn ps={ps}",
                             scale=oracle.csr_f64(1, np.abs(Xn), rp, ci, degrees.numpy()))
            # an output view that is only 4-byte aligned
            buf = torch.full((n * dim + 1,), float("nan"), device="cuda")
            ou = buf[1:].view(n, dim)
            _lib.sag(Xd, rpd, cid, degd, ppd, p2nd, ps, 32, 4, out=ou)
            assert_close_f64(ou.cpu().numpy(), ref, what=f"sparse zero-fill, unaligned out, ps={ps}")
            assert torch.isnan(buf[0])
        # a caller-made partition with groups that hold no edge: one on a row without edges, one in the middle of a
        # row that has edges, one behind the last group
        pp, p2n = (t.numpy() for t in _lib.build_part(4, rpt))
        k = int(np.searchsorted(p2n, 50)) + 2                         # inside the hub's groups
        pp2 = np.concatenate([pp[:1], pp[:1], pp[1:k + 1], pp[k:k + 1], pp[k + 1:], pp[-1:]]).astype(np.int32)
        p2n2 = np.concatenate([[1], p2n[:k], [50], p2n[k:], [n - 2]]).astype(np.int32)
        assert np.all(np.diff(p2n2) >= 0) and np.all(np.diff(pp2) >= 0) and len(pp2) == len(p2n2) + 1
        Xd, rpd, cid, degd, ppd, p2nd = dev(X, rpt, cit, degrees, torch.from_numpy(pp2), torch.from_numpy(p2n2))
        out = torch.full((n, dim), float("nan"), device="cuda")
        _lib.sag(Xd, rpd, cid, degd, ppd, p2nd, 4, 32, 4, out=out)
        assert_close_f64(out.cpu().numpy(), ref, what="sparse zero-fill, groups without edges")
        # groups out of row order: the validation finds it, the whole output is cleared behind the sparse pass
        perm = rng.permutation(len(p2n))
        segs = [ci[pp[j]:pp[j + 1]] for j in perm]
        ci3 = np.concatenate(segs).astype(np.int32)
        pp3 = np.concatenate([[0], np.cumsum([len(s) for s in segs])]).astype(np.int32)
        p2n3 = p2n[perm].astype(np.int32)
        Xd, rpd, cid, degd, ppd, p2nd = dev(X, rpt, torch.from_numpy(ci3), degrees, torch.from_numpy(pp3), torch.from_numpy(p2n3))
        out = torch.full((n, dim), float("nan"), device="cuda")
        _lib.sag(Xd, rpd, cid, degd, ppd, p2nd, 4, 32, 4, out=out)
        assert_close_f64(out.cpu().numpy(), ref, what="sparse zero-fill, shuffled groups")
    finally:
        _lib.reset_tuning()


@pytest.mark.parametrize("spacing,tail", [(5000, 0), (40, 450000), (1, 590000)])
def test_sparse_zero_fill_hands_long_runs_of_empty_rows_to_the_whole_grid(spacing, tail):
    """Rows without edges in runs of >= 1 MiB (trailing isolated / padded rows of a shard, or many long gaps: more than
    the 63 a call's list holds) are cleared by a grid-wide pass behind the sparse prologue, not by the one wavefront
    that finds the gap; every element of a NaN-poisoned output must still be written."""
    n, dim, ps = 600000, 64, 8
    rows_with_edges = torch.arange(0, n - tail, spacing)
    gen = torch.Generator().manual_seed(spacing)
    deg = torch.randint(1, 20, (rows_with_edges.numel(),), generator=gen)
    src = torch.repeat_interleave(rows_with_edges, deg)
    dst = torch.randint(0, n, (int(deg.sum()),), generator=gen)
    g = graph.graph_from_edges(src, dst, n)
    pp, p2n = _lib.build_part(ps, g.row_pointers)
    X = torch.randn(n, dim, generator=gen)
    Xd, rp, ci, degd, ppd, p2nd = dev(X, g.row_pointers, g.column_index, g.degrees, pp, p2n)
    try:
        _lib.set_tuning(zero_fill=1, column_phases=1)
        y = _lib.sag(Xd, rp, ci, degd, ppd, p2nd, ps, 32, 4)             # fresh output: NaN-poisoned by the test config
        yi = _lib.agg_gin(Xd, rp, ci, 0.5, ppd, p2nd, ps, 32, 4)
    finally:
        _lib.reset_tuning()
    ref = oracle.csr_f64(0, X.numpy(), g.row_pointers.numpy(), g.column_index.numpy())
    assert_close_f64(y.cpu().numpy(), ref, what=f"long gaps spacing={spacing} tail={tail}")
    assert_close_f64(yi.cpu().numpy(), 0.5 * ref, what="long gaps gin")


def test_invalid_arguments_are_reported():
    g, X, pp, p2n = make_case(10, 40, 8, 4, seed=1)
    Xd, rp, ci, deg, ppd, p2nd = dev(X, g.row_pointers, g.column_index, g.degrees, pp, p2n)
    with pytest.raises(_lib.GnnaError):
        _lib.sag(Xd, rp, ci, deg, ppd, p2nd, 0, 32, 4)
    with pytest.raises(_lib.GnnaError):
        _lib.sag(Xd, rp, ci, deg, ppd, p2nd, 4, 32, 4, out=Xd)
    with pytest.raises(_lib.GnnaError):
        _lib.sag(X, rp, ci, deg, ppd, p2nd, 4, 32, 4)   # CPU tensor: no CPU path
    # more destination rows than a call can carry: refused before anything is touched
    L = _lib.load()
    rc = L.gnna_sag_f32(Xd.data_ptr(), rp.data_ptr(), ci.data_ptr(), deg.data_ptr(), ppd.data_ptr(), p2nd.data_ptr(),
                        Xd.data_ptr() + 4096, 1 << 29, 8, p2nd.numel(), 4, 32, 4, None)
    assert rc == -3 and b"536870911" in L.gnna_last_error()


def test_full_size_reddit_like_properties():
    """BASELINE config 3 at full size: size-independent properties instead of the oracle.
    (1) X = ones -> exact row-nnz counts; (2) linearity: A(aX + bZ) = a AX + b AZ;
    (3) a 2000-row sample against the fp64 CSR formula."""
    g = graph.make_config_graph("reddit-like", device="cuda")
    n, D = g.num_nodes, 64
    pp, p2n = _lib.build_part(32, g.row_pointers.cpu())
    ppd, p2nd = pp.cuda(), p2n.cuda()
    ones = torch.ones(n, D, device="cuda")
    y1 = _lib.sag(ones, g.row_pointers, g.column_index, g.degrees, ppd, p2nd, 32, 32, 4)
    cnt = (g.row_pointers[1:] - g.row_pointers[:-1]).float()
    assert torch.equal(y1, cnt[:, None].expand(-1, D))
    gen = torch.Generator(device="cuda").manual_seed(3)
    X = torch.randn(n, D, device="cuda", generator=gen)
    Z = torch.randn(n, D, device="cuda", generator=gen)
    yx = _lib.sag(X, g.row_pointers, g.column_index, g.degrees, ppd, p2nd, 32, 32, 4)
    yz = _lib.sag(Z, g.row_pointers, g.column_index, g.degrees, ppd, p2nd, 32, 32, 4)
    yl = _lib.sag(2.0 * X - 0.5 * Z, g.row_pointers, g.column_index, g.degrees, ppd, p2nd, 32, 32, 4)
    ref = 2.0 * yx.double() - 0.5 * yz.double()
    err = (yl.double() - ref).abs()
    # both sides carry fp32 summation error proportional to the sum of |terms|: A (2 |X| + 0.5 |Z|)
    scale = _lib.sag(2.0 * X.abs() + 0.5 * Z.abs(), g.row_pointers, g.column_index, g.degrees, ppd, p2nd, 32, 32, 4).double()
    assert bool((err <= 1e-4 * scale.clamp(min=1.0)).all()), float((err / scale.clamp(min=1.0)).max())
    # sampled rows against the fp64 formula
    rows = torch.randperm(n, generator=torch.Generator().manual_seed(9))[:2000]
    rp_c, ci_c, X_c = g.row_pointers.cpu(), g.column_index.cpu(), X.cpu().double()
    for r in rows[:200].tolist():
        nb = ci_c[rp_c[r]:rp_c[r + 1]].long()
        want = X_c[nb].sum(0)
        got = yx[r].cpu().double()
        assert bool(((got - want).abs() <= 1e-4 * want.abs().clamp(min=1.0)).all()), r


def _check_full_size(g, X, y, rows, what):
    """`rows` sampled rows of y = A X against the fp64 gather-sum: |err| <= 1e-4 * max(1, sum |x_j|)."""
    for r in rows:
        b, e = int(g.row_pointers[r]), int(g.row_pointers[r + 1])
        xs = X[g.column_index[b:e].long()].double()
        err = (y[r].double() - xs.sum(0)).abs()
        assert bool((err <= 1e-4 * xs.abs().sum(0).clamp(min=1.0)).all()), (what, r, float(err.max()))


def test_full_size_reddit_like_through_the_auto_path():
    """BASELINE config 3 exactly as bench.py times it: Decider in auto mode (partSize 128, scheduler knobs), the
    library's own sliced schedule (several phases, one launch).  X = ones -> exact row nnz; sampled rows of a
    randn aggregation vs fp64; and the drop-in call sequence (no Decider, no hints) takes the same schedule."""
    if _lib.get_tuning()["column_phases"] != 0:
        pytest.skip("GNNA_TUNE forces the schedule: the automatic choice is not under test")
    from gnnadvisor_osdi21_amd.decider import inputProperty
    g = graph.make_config_graph("reddit-like", device="cuda")
    n, D = g.num_nodes, 64

    class DS:
        num_nodes, avg_degree, avg_edgeSpan, num_features, reorder_flag = g.num_nodes, g.avg_degree, g.avg_edgeSpan, 602, False

        def rabbit_reorder(self):
            pass
    info = inputProperty(None, None, None, 32, 32, 4, 100, hiddenDim=D, dataset_obj=DS(), manual_mode=False)
    info.decider()
    try:
        info.apply_tuning()
        ps = info.partSize
        assert ps == 128                       # rows of ~490 edges (round 4: 64 until then)
        pp, p2n = _lib.build_part(ps, g.row_pointers.cpu())
        ppd, p2nd = pp.cuda(), p2n.cuda()
        run = lambda x: _lib.sag(x, g.row_pointers, g.column_index, g.degrees, ppd, p2nd, ps, 32, 4)
        cnt = (g.row_pointers[1:] - g.row_pointers[:-1]).float()
        y1 = run(torch.ones(n, D, device="cuda"))
        phases = _lib.last_num_phases()
        assert phases >= 4, phases                               # 59.6 MB of X, scattered ids: sliced
        assert torch.equal(y1, cnt[:, None].expand(-1, D))
        X = torch.randn(n, D, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
        y = run(X)
        rows = [int(torch.argmax(cnt))] + torch.randint(0, n, (255,), generator=torch.Generator().manual_seed(5)).tolist()
        _check_full_size(g, X, y, rows, "auto path")
    finally:
        _lib.reset_tuning()
    # the reference's own call sequence: nothing but build_part and SAG with its manual knobs (32, 32, 4)
    pp32, p2n32 = _lib.build_part(32, g.row_pointers.cpu())
    y32 = _lib.sag(X, g.row_pointers, g.column_index, g.degrees, pp32.cuda(), p2n32.cuda(), 32, 32, 4)
    # (sliced like the auto path; the count may differ: the sweep kernel the library picks at this width keeps a work item
    # -- 64 groups of one phase -- at >= ~256 edges, which halves the phases at partSize 32)
    assert 4 <= _lib.last_num_phases() <= phases
    _check_full_size(g, X, y32, rows[:64], "drop-in path")


def test_config5_papers100M_like_shard():
    """BASELINE config 5, one GPU's share: 1/8 of a papers100M-like graph (13.9 M rows, ~2e8 edges), D = 128
    (rect entry, 7.1 GB of features: 64-bit row offsets), schedule chosen by the library.  X = ones -> exact
    row nnz; sampled rows vs fp64; GCN-weighted form on sampled rows."""
    g = graph.make_config_graph("papers100M-like", device="cuda", scale=0.125)
    n, D, ps = g.num_nodes, 128, 16
    assert n * D * 4 > 2 ** 32
    pp, p2n = _lib.build_part(ps, g.row_pointers.cpu())
    ppd, p2nd = pp.cuda(), p2n.cuda()
    cnt = (g.row_pointers[1:] - g.row_pointers[:-1]).float()
    X = torch.ones(n, D, device="cuda")
    y = _lib.agg_rect(_lib.MODE_SAG, X, g.column_index, ppd, p2nd, n, ps)
    assert torch.equal(y, cnt[:, None].expand(-1, D))
    del y
    X.normal_(generator=torch.Generator(device="cuda").manual_seed(11))
    y = _lib.agg_rect(_lib.MODE_SAG, X, g.column_index, ppd, p2nd, n, ps)
    rows = [int(torch.argmax(cnt)), n - 1, 0] + torch.randint(0, n, (200,), generator=torch.Generator().manual_seed(6)).tolist()
    _check_full_size(g, X, y, rows, "papers100M-like shard")
    try:
        _lib.set_tuning(column_phases=8)                        # the sliced schedule on 64-bit offsets too
        y8 = _lib.agg_rect(_lib.MODE_SAG, X, g.column_index, ppd, p2nd, n, ps)
        assert _lib.last_num_phases() == 8
        _check_full_size(g, X, y8, rows[:64], "papers100M-like shard, 8 phases")
    finally:
        _lib.reset_tuning()
    del y8
    yg = _lib.agg_rect(_lib.MODE_GCN, X, g.column_index, ppd, p2nd, n, ps, degrees_out=g.degrees, degrees_in=g.degrees, out=y)
    for r in rows[:50]:
        b, e = int(g.row_pointers[r]), int(g.row_pointers[r + 1])
        nb = g.column_index[b:e].long()
        coef = (g.degrees[r].double() * g.degrees[nb].double())[:, None]
        ref = (coef * X[nb].double()).sum(0)
        scale = (coef * X[nb].double().abs()).sum(0).clamp(min=1.0)
        assert bool(((yg[r].double() - ref).abs() <= 1e-4 * scale).all()), r


@pytest.mark.parametrize("phases", [2, 3, 8, 16, 24, 32])
@pytest.mark.parametrize("partSize,dim", [(32, 64), (7, 16), (64, 100), (100, 8), (32, 257), (8, 602), (700, 64), (2000, 16)])
def test_column_phased_schedule_matches_single_pass(phases, partSize, dim):
    """The column-phased schedule (X gathered in `phases` source-id ranges, one launch each)
    must give the same answer as the single pass, in every mode.  (partSize 700 / 2000: groups of more than 255
    edges -- the sliced plan's cumulative byte counts saturate there, which may only cost locality.)"""
    g, X, pp, p2n = make_case(1500, 90000, dim, partSize, seed=phases * 100 + dim, kind="powerlaw")
    try:
        _lib.set_tuning(column_phases=phases)
        check_all_modes(g, X, pp, p2n, partSize, what=f"phases={phases} ps={partSize} dim={dim}")
        # exactness on the reference's known-answer input is kept by the phased schedule too
        ones = torch.ones(g.num_nodes, dim)
        ys, _, _ = run_all_modes(g, ones, pp, p2n, partSize)
        want = (g.row_pointers[1:] - g.row_pointers[:-1]).float()[:, None].expand(-1, dim).numpy()
        assert np.array_equal(ys, want)
    finally:
        _lib.reset_tuning()


def test_column_phases_stay_correct_on_unsorted_columns():
    """Column ids shuffled inside every row (legal for the reference kernels, never produced by
    its loader): every edge is still consumed exactly once, only in a less local phase."""
    g, X, pp, p2n = make_case(600, 30000, 64, 16, seed=77, kind="powerlaw")
    ci = g.column_index.clone()
    gen = torch.Generator().manual_seed(1)
    rp = g.row_pointers.tolist()
    for r in range(g.num_nodes):
        seg = ci[rp[r]:rp[r + 1]]
        ci[rp[r]:rp[r + 1]] = seg[torch.randperm(seg.numel(), generator=gen)]
    g2 = graph.CSRGraph(g.num_nodes, g.row_pointers, ci, g.degrees, g.num_edges_raw, g.avg_degree, g.avg_edgeSpan)
    try:
        _lib.set_tuning(column_phases=4)
        check_all_modes(g2, X, pp, p2n, 16, what="unsorted columns, 4 phases")
    finally:
        _lib.reset_tuning()


def test_column_phases_with_empty_rows_hub_and_rect_shapes():
    n = 4000
    src = torch.cat([torch.zeros(n - 1, dtype=torch.int64), torch.arange(1, n)])
    dst = torch.cat([torch.arange(1, n), torch.zeros(n - 1, dtype=torch.int64)])
    gg = graph.graph_from_edges(src, dst, n + 50)            # hub + 50 isolated rows at the end
    pp, p2n = _lib.build_part(8, gg.row_pointers)
    X = torch.randn(n + 50, 32, generator=torch.Generator().manual_seed(4))
    try:
        _lib.set_tuning(column_phases=5)
        check_all_modes(gg, X, pp, p2n, 8, what="hub, 5 phases")
    finally:
        _lib.reset_tuning()


def test_wide_offsets_when_features_exceed_4gib():
    """papers100M-scale feature matrices (BASELINE config 5: D = 128) exceed 4 GiB: the kernel
    must switch to 64-bit row offsets.  9.0 M x 128 fp32 = 4.6 GB of X, few edges, rows picked
    on both sides of the 2^32-byte boundary."""
    n_in, D, n_out = 9_000_000, 128, 4096
    gen = torch.Generator(device="cuda").manual_seed(11)
    X = torch.randn(n_in, D, device="cuda", generator=gen)
    cols = torch.randint(0, n_in, (n_out, 6), device="cuda", generator=gen)
    cols[:, 0] = n_in - 1 - torch.arange(n_out, device="cuda")          # far beyond the 4 GiB mark
    cols[:, 1] = (2 ** 32 // (D * 4)) + torch.arange(n_out, device="cuda") - n_out // 2   # straddling it
    cols, _ = torch.sort(cols, dim=1)
    rp = (torch.arange(n_out + 1, device="cuda") * 6).to(torch.int32)
    ci = cols.reshape(-1).to(torch.int32).contiguous()
    pp, p2n = _lib.build_part(4, rp.cpu())
    y = _lib.agg_rect(_lib.MODE_SAG, X, ci, pp.cuda(), p2n.cuda(), n_out, 4)
    want = X[cols.reshape(-1)].reshape(n_out, 6, D).double().sum(1)
    assert bool(((y.double() - want).abs() <= 1e-4 * want.abs().clamp(min=1.0)).all())
    del X


def test_full_size_products_like_gin_widths():
    """BASELINE config 4 (ogbn-products-like, GIN): layer-1 aggregation at D = F = 100 and hidden
    layers at D = 64, epsilon = 0.5; size-independent checks (ones -> eps * row nnz exactly,
    sampled rows vs fp64)."""
    g = graph.make_config_graph("products-like", device="cuda")
    n = g.num_nodes
    pp, p2n = _lib.build_part(32, g.row_pointers.cpu())
    ppd, p2nd = pp.cuda(), p2n.cuda()
    cnt = (g.row_pointers[1:] - g.row_pointers[:-1]).float()
    rp_c, ci_c = g.row_pointers.cpu(), g.column_index.cpu()
    for D in (100, 64):
        ones = torch.ones(n, D, device="cuda")
        y = _lib.agg_gin(ones, g.row_pointers, g.column_index, 0.5, ppd, p2nd, 32, 32, 4)
        assert torch.equal(y, (0.5 * cnt)[:, None].expand(-1, D))
        del ones, y
        X = torch.randn(n, D, device="cuda", generator=torch.Generator(device="cuda").manual_seed(D))
        y = _lib.agg_gin(X, g.row_pointers, g.column_index, 0.5, ppd, p2nd, 32, 32, 4)
        rows = torch.randperm(n, generator=torch.Generator().manual_seed(1))[:300].tolist()
        Xc = X.cpu().double()
        for r in rows:
            want = 0.5 * Xc[ci_c[rp_c[r]:rp_c[r + 1]].long()].sum(0)
            got = y[r].cpu().double()
            assert bool(((got - want).abs() <= 1e-4 * want.abs().clamp(min=1.0)).all()), (D, r)
        del X, y, Xc


def test_randomised_configurations():
    """Seeded sweep over graph shapes, widths, partition sizes and scheduler knobs (each case
    checked in all three modes against the fp64 formulas and the fp32 oracle)."""
    rng = np.random.default_rng(int(os.environ.get("GNNA_TEST_SEED", "2024")))
    try:
        for k in range(int(os.environ.get("GNNA_TEST_CASES", "40"))):
            n = int(rng.integers(1, 1500))
            e = int(rng.integers(0, 40 * n + 1))
            dim = int(rng.choice([1, 2, 3, 5, 8, 12, 16, 24, 33, 64, 65, 96, 128, 200, 257]))
            ps = int(rng.choice([1, 2, 5, 8, 16, 32, 33, 64, 128, 300]))
            kind = "powerlaw" if rng.random() < 0.5 and n > 10 and e > 20 else "uniform"
            _lib.set_tuning(int(rng.integers(1, 64)), int(rng.choice([4, 8, 16])), int(rng.choice([0, 0, 1, 3])),
                            int(rng.integers(0, 2)), 0, int(rng.choice([1, 1, 2, 5, 16])),
                            gcn_prescale=int(rng.choice([0, 1, 2])))
            g, X, pp, p2n = make_case(n, e, dim, ps, seed=1000 + k, kind=kind)
            check_all_modes(g, X, pp, p2n, ps, eps=float(rng.uniform(-1, 2)),
                            what=f"case {k}: n={n} e={e} dim={dim} ps={ps} {kind} tuning={_lib.get_tuning()}")
    finally:
        _lib.reset_tuning()


def test_sliced_schedule_is_chosen_from_the_partition_itself():
    """column_phases = 0 with the streaming kernel: no hints from anybody (the drop-in caller's situation).  The
    library counts, once per graph, how the column ids of the neighbor-groups spread over 16 source slices and
    picks the number of phases itself: several for a randomly labelled high-degree graph, one when the rows
    are already local (community order), one when the rows are too short to be worth slicing."""
    if _lib.get_tuning()["column_phases"] != 0 or _lib.get_tuning()["deterministic"] != 0:
        pytest.skip("GNNA_TUNE forces the schedule: the automatic choice is not under test")
    seen = {}
    for name, D, g in (("random", 256, graph.make_config_graph("reddit-like", device="cuda", scale=0.25)),
                       # local: every edge within +-4096 ids, a 2 MB window of 64-float rows
                       ("local", 64, graph.make_config_graph("reddit-like", device="cuda", scale=0.5, locality=1.0)),
                       ("sparse", 256, graph.uniform_graph(60000, 360000, seed=5, device="cuda"))):
        pp, p2n = _lib.build_part(64, g.row_pointers.cpu())
        ppd, p2nd = pp.cuda(), p2n.cuda()
        X = torch.randn(g.num_nodes, D, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
        y = _lib.sag(X, g.row_pointers, g.column_index, g.degrees, ppd, p2nd, 64, 32, 4)
        seen[name] = _lib.last_num_phases()
        _lib.sag(X, g.row_pointers, g.column_index, g.degrees, ppd, p2nd, 64, 32, 4)
        assert _lib.last_num_phases() == seen[name]            # second call: same choice, from the cached plan
        try:
            _lib.set_tuning(column_phases=1)
            y1 = _lib.sag(X, g.row_pointers, g.column_index, g.degrees, ppd, p2nd, 64, 32, 4)
        finally:
            _lib.reset_tuning()
        scale = _lib.sag(X.abs(), g.row_pointers, g.column_index, g.degrees, ppd, p2nd, 64, 32, 4).double()
        assert bool(((y.double() - y1.double()).abs() <= 1e-5 * scale.clamp(min=1.0)).all()), name
    assert seen["random"] >= 4 and seen["local"] == 1 and seen["sparse"] == 1, seen


def test_stale_slice_plan_costs_locality_not_correctness():
    """The library caches the slice counts of a partition by the device addresses of its arrays.  Overwriting the
    column ids IN PLACE (same addresses, same partition, different graph) leaves a stale plan behind: the sliced
    schedule must still consume every edge exactly once (the stored counts are prefixes, so the phases partition
    each group's positions whatever they hold) -- also with unsorted ids."""
    g, X, pp, p2n = make_case(3000, 240000, 64, 16, seed=77, kind="powerlaw")
    Xd, rp, ci, deg, ppd, p2nd = dev(X, g.row_pointers, g.column_index, g.degrees, pp, p2n)
    for forced in (8, 32):
      ci.copy_(g.column_index)
      try:
        _lib.set_tuning(column_phases=forced)
        y0 = _lib.sag(Xd, rp, ci, deg, ppd, p2nd, 16, 32, 4)
        assert _lib.last_num_phases() == forced
        assert_close_f64(y0.cpu().numpy(), oracle.csr_f64(0, X.numpy(), g.row_pointers.numpy(), g.column_index.numpy()),
                         what="fresh plan")
        gen = torch.Generator().manual_seed(5)
        for what, new_ci in (("other sorted ids", torch.sort(torch.randint(0, g.num_nodes, (g.nnz,), generator=gen)).values),
                             ("unsorted ids", torch.randint(0, g.num_nodes, (g.nnz,), generator=gen)),
                             ("all ids in the last slice", torch.full((g.nnz,), g.num_nodes - 1))):
            ci.copy_(new_ci.to(torch.int32))                       # in place: the cached plan now describes another graph
            y = _lib.sag(Xd, rp, ci, deg, ppd, p2nd, 16, 32, 4)
            assert _lib.last_num_phases() == forced
            ref = oracle.csr_f64(0, X.numpy(), g.row_pointers.numpy(), new_ci.to(torch.int32).numpy())
            scale = oracle.csr_f64(0, X.abs().numpy(), g.row_pointers.numpy(), new_ci.to(torch.int32).numpy())
            assert_close_f64(y.cpu().numpy(), ref, what=what, scale=scale)
      finally:
        _lib.reset_tuning()


def test_a_scattered_ids_hint_overrides_the_locality_test():
    """The library measures a partition's locality itself (share of the edges near the diagonal) and keeps a
    locality-ordered graph single pass; the Decider's "scattered ids" hint (gnna_tuning.nonlocal_ids / the per-graph form)
    overrides that test.  Results do not depend on the schedule."""
    if _lib.get_tuning()["column_phases"] != 0:
        pytest.skip("GNNA_TUNE forces a phase count: the automatic choice is not under test")
    g = graph.make_config_graph("reddit-like", device="cuda", scale=0.25, locality=0.97)
    pp, p2n = _lib.build_part(64, g.row_pointers.cpu())
    ppd, p2nd = pp.cuda(), p2n.cuda()
    X = torch.randn(g.num_nodes, 64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    try:
        y1 = _lib.sag(X, g.row_pointers, g.column_index, g.degrees, ppd, p2nd, 64, 32, 4)
        assert _lib.last_num_phases() == 1                     # ids near the diagonal: the L2 keeps the window anyway
        _lib.set_tuning(nonlocal_ids=1)
        y2 = _lib.sag(X, g.row_pointers, g.column_index, g.degrees, ppd, p2nd, 64, 32, 4)
        assert _lib.last_num_phases() >= 2                     # told otherwise: X = 14.9 MB is sliced
    finally:
        _lib.reset_tuning()
    scale = _lib.sag(X.abs(), g.row_pointers, g.column_index, g.degrees, ppd, p2nd, 64, 32, 4).double()
    err = (y1.double() - y2.double()).abs()
    assert bool((err <= 1e-5 * scale.clamp(min=1.0)).all())   # two fp32 summation orders


@pytest.mark.parametrize("prescale", [1, 2])
@pytest.mark.parametrize("dim,partSize,phases", [(64, 32, 1), (7, 4, 1), (100, 16, 3), (256, 64, 1)])
def test_gcn_prescaled_and_per_edge_forms_agree_with_the_oracle(prescale, dim, partSize, phases):
    """gnna_agg_gcn_f32 has two arithmetic forms: per-edge coefficients round(deg_i*deg_j)*x as the
    reference computes them (gcn_prescale=2), and rows pre-scaled by deg_j with one multiply by
    deg_i at the flush (gcn_prescale=1).  Both must satisfy the same bound against the oracle."""
    g, X, pp, p2n = make_case(1200, 60000, dim, partSize, seed=dim * 3 + prescale, kind="powerlaw")
    try:
        _lib.set_tuning(column_phases=phases, gcn_prescale=prescale)
        check_all_modes(g, X, pp, p2n, partSize, what=f"prescale={prescale} dim={dim} ps={partSize} ph={phases}")
    finally:
        _lib.reset_tuning()


def test_concurrent_streams_do_not_share_scratch():
    """Two graphs aggregated back-to-back on two HIP streams with the column-phased schedule
    (per-stream cursor workspace, per-call validation flag): results must not interfere."""
    cases = []
    for seed, n, e in ((1, 4000, 300000), (2, 2500, 150000)):
        g, X, pp, p2n = make_case(n, e, 64, 16, seed=seed, kind="powerlaw")
        ref = oracle.csr_f64(0, X.numpy(), g.row_pointers.numpy(), g.column_index.numpy())
        cases.append((dev(X, g.row_pointers, g.column_index, g.degrees, pp, p2n), ref))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [[], []]
    try:
        _lib.set_tuning(column_phases=4)
        torch.cuda.synchronize()
        for rep in range(6):
            for k in (0, 1):
                with torch.cuda.stream(streams[k]):
                    a, _ = cases[k]
                    outs[k].append(_lib.sag(*a, 16, 32, 4))
        torch.cuda.synchronize()
    finally:
        _lib.reset_tuning()
    for k in (0, 1):
        for y in outs[k]:
            assert_close_f64(y.cpu().numpy(), cases[k][1], what=f"stream {k}")


@pytest.mark.parametrize("dim", [4, 5, 7, 16, 32, 41, 64, 100, 128, 256, 300])
@pytest.mark.parametrize("partSize", [1, 32])
def test_sddmm_extension_matches_dense_formula(dim, partSize):
    """Build-defined SDDMM over the neighbor-group partition (not in the reference; parity =
    the dense fp64 formula): edge_out[e] = <A[row(e)], B[col(e)]>."""
    g, _, pp, p2n = make_case(700, 20000, dim, partSize, seed=dim + partSize, kind="powerlaw")
    gen = torch.Generator().manual_seed(dim)
    A = torch.randn(g.num_nodes, dim, generator=gen)
    B = torch.randn(g.num_nodes, dim, generator=gen)
    out = _lib.sddmm(A.cuda(), B.cuda(), g.column_index.cuda(), pp.cuda(), p2n.cuda(), partSize)
    ref = oracle.np_sddmm(A.numpy(), B.numpy(), g.row_pointers.numpy(), g.column_index.numpy())
    rows = np.repeat(np.arange(g.num_nodes), np.diff(g.row_pointers.numpy()))
    scale = np.einsum("ed,ed->e", np.abs(A.numpy().astype(np.float64))[rows],
                      np.abs(B.numpy().astype(np.float64))[g.column_index.numpy()])
    assert_close_f64(out.cpu().numpy(), ref, what=f"sddmm dim={dim} ps={partSize}", scale=scale, rtol=1e-5)
    # rectangular: destination and source sides of different height
    A2 = A[:300].contiguous()
    rp2 = g.row_pointers[:301]
    pp2, p2n2 = _lib.build_part(partSize, rp2.contiguous())
    ci2 = g.column_index[: int(rp2[-1])].contiguous()
    out2 = _lib.sddmm(A2.cuda(), B.cuda(), ci2.cuda(), pp2.cuda(), p2n2.cuda(), partSize)
    assert_close_f64(out2.cpu().numpy(), ref[: int(rp2[-1])], what="sddmm rect", scale=scale[: int(rp2[-1])], rtol=1e-5)


@pytest.mark.parametrize("dim,heads", [(16, 4), (64, 2), (5, 3), (41, 2), (128, 2)])
def test_sddmm_leading_dimensions_heads_of_a_wider_matrix(dim, heads):
    """gnna_sddmm_ld_f32: both sides are column blocks of [N, heads * dim] matrices (one attention head each), and the
    gathered side in the gapped layout gnna_preferred_ld names; the other columns are poisoned and must not be read."""
    g = graph.powerlaw_graph(900, 40000, 300, seed=dim * heads)
    ps = 16
    pp, p2n = _lib.build_part(ps, g.row_pointers)
    gen = torch.Generator().manual_seed(dim + heads)
    Aw = torch.randn(g.num_nodes, heads * dim, generator=gen); Bw = torch.randn(g.num_nodes, heads * dim, generator=gen)
    rows = np.repeat(np.arange(g.num_nodes), np.diff(g.row_pointers.numpy()))
    ci, ppd, p2nd = g.column_index.cuda(), pp.cuda(), p2n.cuda()
    Ad, Bd = Aw.cuda(), Bw.cuda()
    for h in range(heads):
        A, B = Aw[:, h * dim:(h + 1) * dim].contiguous(), Bw[:, h * dim:(h + 1) * dim].contiguous()
        ref = oracle.np_sddmm(A.numpy(), B.numpy(), g.row_pointers.numpy(), g.column_index.numpy())
        scale = np.einsum("ed,ed->e", np.abs(A.numpy().astype(np.float64))[rows],
                          np.abs(B.numpy().astype(np.float64))[g.column_index.numpy()])
        out = _lib.sddmm(Ad[:, h * dim:(h + 1) * dim], Bd[:, h * dim:(h + 1) * dim], ci, ppd, p2nd, ps)
        assert_close_f64(out.cpu().numpy(), ref, what=f"sddmm head {h} of {heads} x {dim}", scale=scale, rtol=1e-5)
    # the gathered side in the library's preferred (gapped) layout, the rest of every row poisoned
    ld = max(_lib.preferred_ld(dim, g.num_nodes, int(g.column_index.numel())), dim + 3)
    Bg = torch.full((g.num_nodes, ld), float("nan"))
    Bg[:, :dim] = Bw[:, :dim]
    Bgd = Bg.cuda()
    A, B = Aw[:, :dim].contiguous(), Bw[:, :dim].contiguous()
    ref = oracle.np_sddmm(A.numpy(), B.numpy(), g.row_pointers.numpy(), g.column_index.numpy())
    scale = np.einsum("ed,ed->e", np.abs(A.numpy().astype(np.float64))[rows], np.abs(B.numpy().astype(np.float64))[g.column_index.numpy()])
    try:
        for phases in (1, 4):
            _lib.set_tuning(column_phases=phases)
            out = _lib.sddmm(Ad[:, :dim], Bgd[:, :dim], ci, ppd, p2nd, ps)
            assert_close_f64(out.cpu().numpy(), ref, what=f"sddmm gapped ld={ld} phases={phases}", scale=scale, rtol=1e-5)
    finally:
        _lib.reset_tuning()
    with pytest.raises(_lib.GnnaError):
        _lib._check(_lib.load().gnna_sddmm_ld_f32(Ad.data_ptr(), dim - 1, Bd.data_ptr(), dim, ci.data_ptr(), ppd.data_ptr(),
                                                  p2nd.data_ptr(), out.data_ptr(), g.num_nodes, g.num_nodes, dim,
                                                  p2nd.numel(), ps, None))


@pytest.mark.parametrize("phases,prescale", [(0, 0), (4, 0), (3, 1)])
def test_hip_graph_capture_and_replay(phases, prescale):
    """The launch path never synchronises and allocates scratch only on first use, so (after one
    warm-up call) a forward can be captured into a HIP graph and replayed on new feature values."""
    g, Xc, ppc, p2nc = make_case(3000, 300000, 64, 32, seed=31, kind="powerlaw")
    rp, ci, deg = g.row_pointers.numpy(), g.column_index.numpy(), g.degrees.numpy()
    X, rpd, cid, degd, pp, p2n = dev(Xc, g.row_pointers, g.column_index, g.degrees, ppc, p2nc)
    out_s, out_g = torch.empty_like(X), torch.empty_like(X)
    try:
        _lib.set_tuning(column_phases=phases, gcn_prescale=prescale)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):                     # warm-up on the capture stream: scratch gets allocated
            _lib.sag(X, rpd, cid, degd, pp, p2n, 32, 32, 4, out=out_s)
            _lib.agg_gcn(X, rpd, cid, degd, pp, p2n, 32, 32, 4, out=out_g)
        side.synchronize()
        hg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(hg, stream=side):
            _lib.sag(X, rpd, cid, degd, pp, p2n, 32, 32, 4, out=out_s)
            _lib.agg_gcn(X, rpd, cid, degd, pp, p2n, 32, 32, 4, out=out_g)
        for seed in (1, 2, 3):
            X.copy_(torch.randn(X.shape, generator=torch.Generator().manual_seed(seed)))
            out_s.fill_(float("nan")); out_g.fill_(float("nan"))
            hg.replay()
            torch.cuda.synchronize()
            Xn = X.cpu().numpy()
            assert_close_f64(out_s.cpu().numpy(), oracle.csr_f64(0, Xn, rp, ci, deg), what=f"graph sag {seed}",
                             scale=oracle.csr_f64(0, np.abs(Xn), rp, ci, deg))
            assert_close_f64(out_g.cpu().numpy(), oracle.csr_f64(1, Xn, rp, ci, deg), what=f"graph gcn {seed}",
                             scale=oracle.csr_f64(1, np.abs(Xn), rp, ci, deg))
    finally:
        _lib.reset_tuning()


@pytest.mark.parametrize("K,dim,partSize,sorted_ids", [(2, 64, 32, True), (4, 64, 16, True), (3, 41, 8, True),
                                                         (4, 64, 32, False), (16, 16, 4, True), (4, 257, 32, True)])
def test_windowed_calls_pipeline_with_arriving_source_rows(K, dim, partSize, sorted_ids):
    """gnna_agg_rect_windows_f32: the source rows arrive window by window (rows that have not arrived
    hold NaN); one call per window, in order, must reproduce the one-shot aggregation in all three
    modes, with and without accumulate.  Shuffled column ids are refused (a window call takes id positions)."""
    g, Xc, ppc, p2nc = make_case(3000, 200000, dim, partSize, seed=K * 100 + dim, kind="powerlaw")
    rp, deg = g.row_pointers.numpy(), g.degrees.numpy()
    ci_t = g.column_index.clone()
    if not sorted_ids:                                     # shuffle the ids inside every row
        gen = torch.Generator().manual_seed(9)
        rows = torch.repeat_interleave(torch.arange(g.num_nodes), (g.row_pointers[1:] - g.row_pointers[:-1]).long())
        order = torch.argsort(rows.double() + torch.rand(ci_t.numel(), generator=gen, dtype=torch.float64) * 0.5)
        ci_t = ci_t[order].contiguous()
    ci = ci_t.numpy()
    n = g.num_nodes
    wrows = (n + K - 1) // K
    X, cid, degd, pp, p2n = dev(Xc, ci_t, g.degrees, ppc, p2nc)
    Xn = Xc.numpy()
    base = torch.randn(n, dim, generator=torch.Generator().manual_seed(4))
    if not sorted_ids:
        out = torch.zeros(n, dim, device="cuda")
        if K == 1:          # one window is the whole aggregation: no order needed
            _lib.agg_rect(0, X, cid, pp, p2n, n, partSize, out=out, windows=(K, 0, 1))
            assert_close_f64(out.cpu().numpy(), oracle.csr_f64(0, Xn, rp, ci, deg), what="one window, shuffled ids",
                             scale=oracle.csr_f64(0, np.abs(Xn), rp, ci, deg))
        else:
            with pytest.raises(_lib.GnnaError, match="not in increasing order"):
                _lib.agg_rect(0, X, cid, pp, p2n, n, partSize, out=out, windows=(K, 0, 1))
        return
    for mode, eps, prescale in ((0, 1.0, 0), (1, 1.0, -1), (1, 1.0, 1), (2, 0.5, 0)):
        ref = oracle.csr_f64(mode, Xn, rp, ci, deg, eps)
        scale = oracle.csr_f64(mode, np.abs(Xn), rp, ci, deg, abs(eps))
        for accumulate in (False, True):
            try:
                _lib.set_tuning(gcn_prescale=prescale)
                live = torch.full_like(X, float("nan"))
                out = base.clone().cuda() if accumulate else torch.full((n, dim), float("nan"), device="cuda")
                for k in range(K):
                    lo, hi = min(k * wrows, n), min((k + 1) * wrows, n)
                    live[lo:hi] = X[lo:hi]
                    _lib.agg_rect(mode, live, cid, pp, p2n, n, partSize, degrees_out=degd, degrees_in=degd,
                                  epsilon=eps, out=out, accumulate=accumulate, windows=(K, k, k + 1))
                assert _lib.last_num_launches() == 1
            finally:
                _lib.reset_tuning()
            want = ref + (base.double().numpy() if accumulate else 0.0)
            assert_close_f64(out.cpu().numpy(), want, what=f"windows K={K} mode={mode} acc={accumulate}",
                             scale=scale + (np.abs(base.numpy()) if accumulate else 0.0))
    # several windows per call (a forced phase count does not apply to window calls)
    try:
        _lib.set_tuning(column_phases=2 * K)
        out = torch.empty(n, dim, device="cuda")
        half = max(1, K // 2)
        _lib.agg_rect(0, X, cid, pp, p2n, n, partSize, out=out, windows=(K, 0, half))
        if half < K:
            _lib.agg_rect(0, X, cid, pp, p2n, n, partSize, out=out, windows=(K, half, K))
    finally:
        _lib.reset_tuning()
    assert_close_f64(out.cpu().numpy(), oracle.csr_f64(0, Xn, rp, ci, deg), what="grouped windows",
                     scale=oracle.csr_f64(0, np.abs(Xn), rp, ci, deg))
    with pytest.raises(_lib.GnnaError):
        _lib.agg_rect(0, X, cid, pp, p2n, n, partSize, out=out, windows=(K, 1, 1))
    with pytest.raises(_lib.GnnaError):
        _lib.agg_rect(0, X, cid, pp, p2n, n, partSize, out=out, windows=(17, 0, 1))
    if K >= 2:
        # stateless between the calls: another aggregation (any schedule, any graph) in the middle of a sequence does
        # not disturb it
        out = torch.full((n, dim), float("nan"), device="cuda")
        _lib.agg_rect(0, X, cid, pp, p2n, n, partSize, out=out, windows=(K, 0, 1))
        try:
            _lib.set_tuning(column_phases=3)
            _lib.sag(X, None, cid, None, pp, p2n, partSize, 32, 4)
        finally:
            _lib.reset_tuning()
        _lib.agg_rect(0, X, cid, pp, p2n, n, partSize, out=out, windows=(K, 1, K))
        assert_close_f64(out.cpu().numpy(), oracle.csr_f64(0, Xn, rp, ci, deg), what="interleaved windows",
                         scale=oracle.csr_f64(0, np.abs(Xn), rp, ci, deg))


@pytest.mark.parametrize("dim", [3, 5, 7, 22, 41, 47, 56, 60, 100, 172])
@pytest.mark.parametrize("pad", [1, 2])
def test_padded_row_staging_does_not_change_results(dim, pad):
    """gnna_tuning.pad_rows: gathering from a staged copy of X with a line-friendly row stride
    (41 -> 48 floats ...) against gathering from X itself -- all modes, phases, accumulate."""
    g, Xc, ppc, p2nc = make_case(1500, 90000, dim, 16, seed=dim, kind="powerlaw")
    rp, ci, deg, Xn = g.row_pointers.numpy(), g.column_index.numpy(), g.degrees.numpy(), Xc.numpy()
    X, rpd, cid, degd, pp, p2n = dev(Xc, g.row_pointers, g.column_index, g.degrees, ppc, p2nc)
    n = g.num_nodes
    try:
        for phases in (1, 3):
            _lib.set_tuning(pad_rows=pad, column_phases=phases)
            for mode, eps, fn in ((0, 1.0, lambda o: _lib.sag(X, rpd, cid, degd, pp, p2n, 16, 32, 4, out=o)),
                                  (1, 1.0, lambda o: _lib.agg_gcn(X, rpd, cid, degd, pp, p2n, 16, 32, 4, out=o)),
                                  (2, 0.5, lambda o: _lib.agg_gin(X, rpd, cid, 0.5, pp, p2n, 16, 32, 4, out=o))):
                out = torch.full((n, dim), float("nan"), device="cuda")
                fn(out)
                assert_close_f64(out.cpu().numpy(), oracle.csr_f64(mode, Xn, rp, ci, deg, eps),
                                 what=f"pad={pad} D={dim} mode={mode} phases={phases}",
                                 scale=oracle.csr_f64(mode, np.abs(Xn), rp, ci, deg, eps))
        base = torch.randn(n, dim, generator=torch.Generator().manual_seed(2))
        out = base.clone().cuda()
        _lib.set_tuning(pad_rows=pad, column_phases=0)
        for k in range(2):
            _lib.agg_rect(1, X, cid, pp, p2n, n, 16, degrees_out=degd, degrees_in=degd, out=out, accumulate=True,
                          windows=(2, k, k + 1))
        assert_close_f64(out.cpu().numpy(), oracle.csr_f64(1, Xn, rp, ci, deg) + base.double().numpy(),
                         what=f"pad={pad} D={dim} windows+accumulate",
                         scale=oracle.csr_f64(1, np.abs(Xn), rp, ci, deg) + np.abs(base.numpy()))
    finally:
        _lib.reset_tuning()


def test_per_graph_hints_are_keyed_by_the_column_index_array():
    """gnna_set_graph_hints: two graphs alive at once get their own schedule; forgetting a graph
    falls back to the process-wide hints; results never depend on the hints."""
    if _lib.get_tuning()["column_phases"] != 0:
        pytest.skip("GNNA_TUNE forces a phase count: the automatic choice is not under test")
    # two locality-ordered graphs: the library keeps them single pass unless it is told that the ids are scattered
    g1 = graph.make_config_graph("reddit-like", device="cuda", scale=0.25, locality=0.97)
    g2 = graph.make_config_graph("reddit-like", device="cuda", scale=0.2, locality=0.97)
    parts = []
    for g in (g1, g2):
        pp, p2n = _lib.build_part(64, g.row_pointers.cpu())
        parts.append((pp.cuda(), p2n.cuda()))
    X1 = torch.randn(g1.num_nodes, 64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    X2 = torch.randn(g2.num_nodes, 64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
    run1 = lambda: _lib.sag(X1, g1.row_pointers, g1.column_index, g1.degrees, *parts[0], 64, 32, 4)
    run2 = lambda: _lib.sag(X2, g2.row_pointers, g2.column_index, g2.degrees, *parts[1], 64, 32, 4)
    try:
        y1 = run1(); assert _lib.last_num_phases() == 1
        _lib.set_graph_hints(g1.column_index, g1.nnz / g1.num_nodes, True)
        y1h = run1(); assert _lib.last_num_phases() >= 2       # 14.9 MB of X, "scattered" ids, degree ~490
        run2(); assert _lib.last_num_phases() == 1             # the other graph is unaffected
        _lib.set_graph_hints(g2.column_index, 30, False)
        _lib.set_tuning(avg_degree=500, nonlocal_ids=1)        # process-wide hints lose against per-graph ones
        run2(); assert _lib.last_num_phases() == 1
        _lib.set_graph_hints(g2.column_index, 0, False)        # forget g2 -> process-wide hints apply
        run2(); assert _lib.last_num_phases() >= 2
        _lib.set_graph_hints(None, 0, False)                   # forget everything
        _lib.reset_tuning()
        run1(); assert _lib.last_num_phases() == 1
    finally:
        _lib.set_graph_hints(None, 0, False)
        _lib.reset_tuning()
    scale = _lib.sag(X1.abs(), g1.row_pointers, g1.column_index, g1.degrees, *parts[0], 64, 32, 4).double()
    assert bool(((y1.double() - y1h.double()).abs() <= 1e-5 * scale.clamp(min=1.0)).all())


@pytest.mark.parametrize("M,K,N", [(1000, 64, 64), (4097, 100, 47), (333, 602, 41), (50000, 16, 22), (17, 3, 5),
                                   (1, 7, 9), (0, 8, 8), (70001, 130, 65), (256, 1, 1)])
def test_weight_gradient_kernel_matches_fp64(M, K, N):
    """gnna_xtg_f32: dW = X^T G (MFMA, fp32) against the fp64 product; tails in all three dimensions."""
    gen = torch.Generator().manual_seed(M + K + N)
    X = torch.randn(M, K, generator=gen); G = torch.randn(M, N, generator=gen)
    got = _lib.xtg(X.cuda(), G.cuda())
    ref = X.double().t() @ G.double()
    scale = X.double().abs().t() @ G.double().abs()
    assert got.shape == (K, N)
    assert_close_f64(got.cpu().numpy(), ref.numpy(), rtol=1e-5, what=f"xtg {M}x{K}x{N}", scale=scale.numpy())
    # deterministic: the same call returns the same bits
    assert torch.equal(got, _lib.xtg(X.cuda(), G.cuda()))


@pytest.mark.parametrize("dim,partSize,phases,sorted_ids", [(64, 32, 3, True), (16, 8, 5, True), (41, 64, 2, True),
                                                            (300, 16, 4, True), (64, 32, 4, False), (64, 1, 16, True),
                                                            (64, 600, 8, True), (16, 600, 16, False)])
def test_sddmm_column_phases_match_single_pass(dim, partSize, phases, sorted_ids):
    """SDDMM with the column-phased schedule (per-run cursors): every edge written exactly once, for
    sorted and shuffled column ids, partitions whose rows span several groups, and empty rows."""
    g = graph.powerlaw_graph(1200, 90000, 600, seed=dim + phases)
    ci_t = g.column_index.clone()
    if not sorted_ids:
        rows = torch.repeat_interleave(torch.arange(g.num_nodes), (g.row_pointers[1:] - g.row_pointers[:-1]).long())
        order = torch.argsort(rows.double() + torch.rand(ci_t.numel(), generator=torch.Generator().manual_seed(1),
                                                         dtype=torch.float64) * 0.5)
        ci_t = ci_t[order].contiguous()
    pp, p2n = _lib.build_part(partSize, g.row_pointers)
    gen = torch.Generator().manual_seed(dim)
    A = torch.randn(g.num_nodes, dim, generator=gen); B = torch.randn(g.num_nodes, dim, generator=gen)
    ref = oracle.np_sddmm(A.numpy(), B.numpy(), g.row_pointers.numpy(), ci_t.numpy())
    rows = np.repeat(np.arange(g.num_nodes), np.diff(g.row_pointers.numpy()))
    scale = np.einsum("ed,ed->e", np.abs(A.numpy().astype(np.float64))[rows], np.abs(B.numpy().astype(np.float64))[ci_t.numpy()])
    try:
        _lib.set_tuning(column_phases=phases)
        out = torch.full((ci_t.numel(),), float("nan"), device="cuda")
        _lib.sddmm(A.cuda(), B.cuda(), ci_t.cuda(), pp.cuda(), p2n.cuda(), partSize, out=out)
    finally:
        _lib.reset_tuning()
    assert_close_f64(out.cpu().numpy(), ref, what=f"sddmm phases={phases}", scale=scale, rtol=1e-5)
