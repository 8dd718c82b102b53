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
