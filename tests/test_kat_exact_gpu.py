"""Exact-arithmetic known-answer tests for the weighted kernels K2-K5 (reference
GNNAdvisor_kernel.cu:324-415 GCN forward, :478-552 GCN backward, :620-689 GIN forward, :749-814 GIN
backward; host glue :280-282, :472-473, :605-616, :710-746).

The reference's CUDA cannot run here and its tree holds no vectors for these kernels, so they are
pinned by construction instead: on a graph whose row degrees are perfect squares the degree norm
``sqrt(max(deg, 1))`` is an exact small integer, the GCN coefficient ``deg_i * deg_j`` (.cu:355,389) is
an integer, and with small-integer features / weights / gradients every product and every partial
sum of forward AND backward is an integer (or a multiple of 0.5 for GIN's epsilon) far below 2**24 --
exactly representable in fp32 whatever the summation order.  The module functions must therefore
reproduce the integer results of the formulas BIT FOR BIT, under every schedule the library has."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from gnnadvisor_osdi21_amd import _lib, load_extension

pytestmark = pytest.mark.gpu

SCHEDULES = {
    "default": dict(),
    "sliced_3": dict(column_phases=3),
    "sliced_16_g5": dict(column_phases=16, groups_per_chunk=5),
    "seven_groups_per_item": dict(groups_per_chunk=7),
    "sliced_5_g64_deterministic": dict(column_phases=5, groups_per_chunk=64, deterministic=1),
    "gcn_per_edge": dict(gcn_prescale=2),
    "gcn_prescaled": dict(gcn_prescale=1, pad_rows=1),
    "one_group_per_item": dict(groups_per_chunk=1),
    "sparse_zero_fill": dict(zero_fill=1),
    "sparse_zero_fill_g3": dict(zero_fill=1, groups_per_chunk=3),
}


def square_degree_graph(n, seed, squares=(0, 1, 4, 9, 16, 25)):
    """CSR whose row degrees are perfect squares (0 counts as degree-norm 1, dataset.py:11-18)."""
    rng = np.random.default_rng(seed)
    deg = rng.choice(squares, size=n, p=None)
    deg[rng.integers(0, n)] = 25
    rp = np.zeros(n + 1, dtype=np.int32)
    rp[1:] = np.cumsum(deg)
    ci = np.concatenate([np.sort(rng.choice(n, size=d, replace=False)) for d in deg] + [np.zeros(0, dtype=np.int64)])
    return rp, ci.astype(np.int32), np.sqrt(np.maximum(deg, 1)).astype(np.float32)


def small_ints(rng, shape, lo, hi):
    return rng.integers(lo, hi + 1, size=shape).astype(np.float32)


@pytest.mark.parametrize("schedule", sorted(SCHEDULES))
@pytest.mark.parametrize("n,fin,fout,ps", [(600, 12, 8, 4), (257, 5, 64, 3), (900, 64, 7, 16)])
def test_weighted_kernels_are_exact_on_square_degree_graphs(schedule, n, fin, fout, ps):
    GNNA = load_extension()
    rp, ci, deg = square_degree_graph(n, seed=n + fin)
    assert np.array_equal(deg, np.round(deg)) and deg.max() == 5.0
    rng = np.random.default_rng(7 * n + fout)
    X, W = small_ints(rng, (n, fin), -3, 3), small_ints(rng, (fin, fout), -2, 2)
    dY = small_ints(rng, (n, fout), -1, 1)
    A = sp.csr_matrix((np.ones(len(ci)), ci, rp), shape=(n, n)).astype(np.float64)
    d64 = deg.astype(np.float64)
    Ahat = sp.diags(d64) @ A @ sp.diags(d64)                  # coefficient deg_i * deg_j, an integer
    X64, W64, dY64 = X.astype(np.float64), W.astype(np.float64), dY.astype(np.float64)

    pp, p2n = GNNA.build_part(ps, torch.from_numpy(rp))
    dev = [torch.from_numpy(a).cuda() for a in (rp, ci, deg)] + [pp.int().cuda(), p2n.int().cuda()]
    rpd, cid, degd, ppd, p2nd = dev
    Xd, Wd, dYd = (torch.from_numpy(a).cuda() for a in (X, W, dY))

    def exact(got, ref64, what):
        assert np.abs(ref64).max() < 2 ** 24, what
        ref = ref64.astype(np.float32)
        assert np.array_equal(ref.astype(np.float64), ref64), what + ": reference is not exactly representable"
        got = got.cpu().numpy()
        assert np.array_equal(got, ref), "%s [%s]: %d elements differ, max |diff| %g" % (
            what, schedule, int((got != ref).sum()), float(np.abs(got - ref).max()))

    _lib.reset_tuning()
    _lib.set_tuning(**SCHEDULES[schedule])
    try:
        # K1 (.cu:186-259) for completeness
        exact(GNNA.SAG(Xd, rpd, cid, degd, ppd, p2nd, ps, 32, 4), A @ X64, "SAG")
        # K2: Y = A_hat (X W)
        exact(GNNA.forward(Xd, Wd, rpd, cid, degd, ppd, p2nd, ps, 32, 4)[0], Ahat @ (X64 @ W64), "GCN forward")
        # K3: G = A_hat dY ; dX = G W^T ; dW = X^T G
        G = Ahat @ dY64
        dX, dW = GNNA.backward(dYd, Xd, Wd, rpd, cid, degd, ppd, p2nd, ps, 32, 4)
        exact(dX, G @ W64.T, "GCN backward d_input")
        exact(dW, X64.T @ G, "GCN backward d_weight")
        # K4: T = eps A X ; Y = T W
        T = 0.5 * (A @ X64)
        Yg, Tg = GNNA.forward_gin(Xd, Wd, rpd, cid, 0.5, ppd, p2nd, ps, 32, 4)
        exact(Tg, T, "GIN aggregated")
        exact(Yg, T @ W64, "GIN forward")
        # K5: dW = T^T dY ; dX = eps A (dY W^T)     (the saved T is passed as X, gnn_conv.py:109,119)
        dXg, dWg = GNNA.backward_gin(dYd, Tg, Wd, rpd, cid, 0.5, ppd, p2nd, ps, 32, 4)
        exact(dWg, T.T @ dY64, "GIN backward d_weight")
        exact(dXg, 0.5 * (A @ (dY64 @ W64.T)), "GIN backward d_input")
    finally:
        _lib.reset_tuning()
