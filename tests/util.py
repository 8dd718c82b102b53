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


def gcn_gin_reference(g, F, W1, W2, wgt, eps=0.5):
    """fp64 reference of  y = GIN(relu(GCN(F)))  = (eps A relu(Ahat (F W1))) W2  with loss = sum(y * wgt) on a SYMMETRIC
    graph (scipy CSR, explicit backward), and the sum-of-|terms| scale of every result: the same network evaluated on
    |F|, |W1|, |W2| with relu' = 1 (A, Ahat, wgt are non-negative), whose values / gradients bound the magnitude sums fp32
    rounding errors are proportional to.  -> {name: (reference, scale)} for out, dF, dW1, dW2 -- compared as
    |got - ref| <= 1e-4 * max(1, scale)."""
    import scipy.sparse as sp
    n = g.num_nodes
    rp, ci, deg = g.row_pointers.numpy(), g.column_index.numpy(), g.degrees.double().numpy()
    A = sp.csr_matrix((np.ones(len(ci)), ci, rp), shape=(n, n))
    Ahat = sp.diags(deg) @ A @ sp.diags(deg)
    w = wgt.detach().double().cpu().numpy()
    res = {}
    for tag in ("ref", "abs"):
        f = (lambda t: t) if tag == "ref" else np.abs
        Fn, W1n, W2n = (f(t.detach().double().cpu().numpy()) for t in (F, W1, W2))
        H1 = Ahat @ (Fn @ W1n)
        mask = (H1 > 0).astype(np.float64) if tag == "ref" else np.ones_like(H1)
        Rl = np.maximum(H1, 0.0) if tag == "ref" else H1
        T = eps * (A @ Rl)
        Y = T @ W2n
        dY = np.broadcast_to(w, Y.shape)
        dW2 = T.T @ dY
        dH1 = (eps * (A.T @ (dY @ W2n.T))) * mask
        G = Ahat.T @ dH1
        res[tag] = dict(out=Y, dF=G @ W1n.T, dW1=Fn.T @ G, dW2=dW2)
    return {k: (res["ref"][k], res["abs"][k]) for k in res["ref"]}
