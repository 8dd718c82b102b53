"""Shared helpers for the parity tests (checker side only)."""
import numpy as np
import torch

import oracle
from gnnadvisor_osdi21_amd import _lib, graph


def make_case(num_nodes, num_edges, dim, partSize, seed, kind="uniform", x="randn"):
    """Seeded CPU graph + features + the product's partition (all numpy/torch CPU)."""
    if kind == "uniform":
        g = graph.uniform_graph(num_nodes, num_edges, seed=seed)
    else:
        g = graph.powerlaw_graph(num_nodes, num_edges, max(1, min(num_nodes - 1, num_edges // 8 + 1)), seed=seed)
    gen = torch.Generator().manual_seed(seed + 1000)
    if x == "ones":
        X = torch.ones(num_nodes, dim)
    else:
        X = torch.randn(num_nodes, dim, generator=gen)
    pp, p2n = _lib.build_part(partSize, g.row_pointers)
    return g, X, pp, p2n


def dev(*ts):
    return [t.cuda() for t in ts]


def assert_close_f64(got, ref64, rtol=1e-4, what="", scale=None):
    """|got - ref| <= rtol * max(1, scale) elementwise (north_star: 1e-4 fp32).

    ``scale`` defaults to |ref|.  For the degree-weighted (GCN) variant the coefficients
    are products of sqrt-degrees, so outputs reach 1e3..1e5 and individual elements can be
    tiny sums of huge terms; there the scale is the sum of |coef * x| (the quantity fp32
    summation error is proportional to), computed by the same fp64 formula on |X|
    (tolerance note)."""
    got = np.asarray(got, dtype=np.float64)
    ref64 = np.asarray(ref64, dtype=np.float64)
    assert got.shape == ref64.shape, (got.shape, ref64.shape)
    err = np.abs(got - ref64)
    tol = rtol * np.maximum(1.0, np.abs(ref64) if scale is None else np.asarray(scale, dtype=np.float64))
    bad = ~(err <= tol)                 # (a NaN in `got` compares false both ways: it must count as off)
    assert not bad.any(), f"{what}: {bad.sum()} / {bad.size} elements off, max err {err.max():.3e}"


def oracle_inputs(g, X, pp, p2n):
    return (X.numpy(), g.column_index.numpy(), pp.numpy(), p2n.numpy())


def dense_adjacency(g, dtype=torch.float64):
    n = g.num_nodes
    A = torch.zeros(n, n, dtype=dtype)
    rows = torch.repeat_interleave(torch.arange(n), (g.row_pointers[1:] - g.row_pointers[:-1]).long())
    A[rows, g.column_index.long()] = 1.0
    return A


def gcn_gin_reference(g, F, W1, W2, wgt, eps=0.5, H1_got=None, amb_tol=1e-5):
    """fp64 reference of  y = GIN(relu(GCN(F)))  = (eps A relu(Ahat (F W1))) W2  with loss = sum(y * wgt) on a SYMMETRIC
    graph (scipy CSR, explicit backward), and the sum-of-|terms| scale of every result: the same network evaluated on
    |F|, |W1|, |W2| with relu' = 1 (A, Ahat, wgt are non-negative), whose values / gradients bound the magnitude sums fp32
    rounding errors are proportional to.  -> {name: (reference, scale)} for out, H1, dF, dW1, dW2 -- compared as
    |got - ref| <= 1e-4 * max(1, scale) -- plus "ambiguous" / "min_ratio" (below).

    ``H1_got``: the layer-1 output of the path under test for ALL nodes ([N, hidden]).  relu' is a step function: a
    pre-activation H1[i, k] that cancels to within fp32 rounding of zero has no defined sign, and the one the fp32 path
    happens to compute (which depends on the summation order, i.e. on the order float atomics land in) decides a whole
    column entry of dH1.  One such flip moves dF in every neighbour row of node i by far more than 1e-4 of the scale
    (a degree-10 node: 9 rows, up to 6e-3) while out / dW1 / dW2 stay inside the bound -- the "two-rank transient" of
    rounds 3-4 (profiles/r5/DESIGN_as_of_round5.md 6).  So for the elements that cancel to |H1_ref| <= amb_tol * (sum of |terms|) -- amb_tol =
    1e-5: ~100 x the fp32 error these sums actually show (<= 1e-7 of the sum of |terms|), 10 x inside the 1e-4 bound H1
    itself is checked with -- the mask follows the sign the path under test computed; everywhere else it is the fp64
    sign.  "ambiguous" = how many elements that concerned (about one per 30,000), "min_ratio" = the smallest
    |H1_ref| / sum of |terms|, "sign_flips" = how many of them the path computed with the other sign than fp64."""
    import scipy.sparse as sp
    n = g.num_nodes
    rp, ci, deg = g.row_pointers.numpy(), g.column_index.numpy(), g.degrees.double().numpy()
    A = sp.csr_matrix((np.ones(len(ci)), ci, rp), shape=(n, n))
    Ahat = sp.diags(deg) @ A @ sp.diags(deg)
    w = wgt.detach().double().cpu().numpy()
    Fd, W1d, W2d = (t.detach().double().cpu().numpy() for t in (F, W1, W2))
    H1_abs = Ahat @ (np.abs(Fd) @ np.abs(W1d))
    H1_ref = Ahat @ (Fd @ W1d)
    mask_ref = H1_ref > 0
    ratio = np.abs(H1_ref) / np.maximum(H1_abs, 1e-300)
    ambiguous = (np.abs(H1_ref) <= amb_tol * H1_abs) & (H1_abs > 0)      # (an isolated node's exact zero is not a cancellation)
    if H1_got is not None:
        got = np.asarray(H1_got.detach().cpu().numpy() if hasattr(H1_got, "detach") else H1_got, dtype=np.float64)
        assert got.shape == H1_ref.shape, (got.shape, H1_ref.shape)
        mask_ref = np.where(ambiguous, got > 0, mask_ref)
    res = {}
    for tag in ("ref", "abs"):
        f = (lambda t: t) if tag == "ref" else np.abs
        Fn, W1n, W2n = f(Fd), f(W1d), f(W2d)
        H1 = H1_ref if tag == "ref" else H1_abs
        mask = mask_ref.astype(np.float64) if tag == "ref" else np.ones_like(H1)
        Rl = np.maximum(H1, 0.0) if tag == "ref" else H1
        T = eps * (A @ Rl)
        Y = T @ W2n
        dY = np.broadcast_to(w, Y.shape)
        dW2 = T.T @ dY
        dH1 = (eps * (A.T @ (dY @ W2n.T))) * mask
        G = Ahat.T @ dH1
        res[tag] = dict(out=Y, H1=H1, dF=G @ W1n.T, dW1=Fn.T @ G, dW2=dW2)
    out = {k: (res["ref"][k], res["abs"][k]) for k in res["ref"]}
    out["ambiguous"] = int(ambiguous.sum())
    out["min_ratio"] = float(ratio.min())
    out["sign_flips"] = int((np.where(ambiguous, got > 0, H1_ref > 0) != (H1_ref > 0)).sum()) if H1_got is not None else 0
    return out
