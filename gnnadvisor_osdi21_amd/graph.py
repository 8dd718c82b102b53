"""Graph inputs for the aggregation path: CSR + degree construction and seeded synthetic
graphs (no dataset ships with the reference and there is no network here).

Reference behaviour restated:

* ``csr_from_edges``  -- GNNAdvisor/dataset.py:108-118: ``coo_matrix(...).tocsr()``, i.e.
  duplicate edges merged and column indices sorted per row; int32 ``row_pointers`` /
  ``column_index``.
* ``degrees_from_rowptr`` -- dataset.py:11-18,121-122: ``sqrt(max(deg, 1))`` as float32.
* ``graph_stats``     -- dataset.py:99-100: ``avg_degree = E_raw / N``,
  ``avg_edgeSpan = mean(|src - dst|)`` (inputs of the Decider).

Everything is torch code that runs on the tensor's device, so a Reddit-scale graph
(1.1e8 edges) is generated and deduplicated on the GPU in well under a second.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch


@dataclass
class CSRGraph:
    num_nodes: int
    row_pointers: torch.Tensor   # int32 [N+1]
    column_index: torch.Tensor   # int32 [nnz]
    degrees: torch.Tensor        # float32 [N] = sqrt(max(deg, 1))
    num_edges_raw: int           # edges before deduplication (dataset.py: self.num_edges)
    avg_degree: float
    avg_edgeSpan: float

    @property
    def nnz(self) -> int:
        return int(self.column_index.numel())

    def to(self, device) -> "CSRGraph":
        return CSRGraph(self.num_nodes, self.row_pointers.to(device), self.column_index.to(device),
                        self.degrees.to(device), self.num_edges_raw, self.avg_degree, self.avg_edgeSpan)


def csr_from_edges(src: torch.Tensor, dst: torch.Tensor, num_nodes: int):
    """Edge list -> (row_pointers int32 [N+1], column_index int32 [nnz]) with duplicates
    merged and per-row sorted columns (scipy coo->csr semantics, dataset.py:110-118)."""
    src = src.to(torch.int64)
    dst = dst.to(torch.int64)
    if src.numel() == 0:
        return (torch.zeros(num_nodes + 1, dtype=torch.int32, device=src.device),
                torch.zeros(0, dtype=torch.int32, device=src.device))
    key = torch.unique(src * num_nodes + dst, sorted=True)
    rows = torch.div(key, num_nodes, rounding_mode="floor")
    cols = key - rows * num_nodes
    counts = torch.bincount(rows, minlength=num_nodes)
    if key.numel() >= 2**31:
        raise ValueError("more than 2^31-1 distinct edges: shard the graph (int32 CSR, see dist.py)")
    rp = torch.zeros(num_nodes + 1, dtype=torch.int64, device=src.device)
    rp[1:] = torch.cumsum(counts, 0)
    return rp.to(torch.int32), cols.to(torch.int32)


def degrees_from_rowptr(row_pointers: torch.Tensor) -> torch.Tensor:
    """dataset.py:121-122: sqrt(max(deg, 1)) as float32."""
    deg = (row_pointers[1:] - row_pointers[:-1]).clamp(min=1).to(torch.float32)
    return torch.sqrt(deg)


def graph_from_edges(src, dst, num_nodes: int) -> CSRGraph:
    rp, ci = csr_from_edges(src, dst, num_nodes)
    e_raw = int(src.numel())
    span = float((src.to(torch.int64) - dst.to(torch.int64)).abs().to(torch.float64).mean()) if e_raw else 0.0
    return CSRGraph(num_nodes, rp, ci, degrees_from_rowptr(rp), e_raw,
                    e_raw / max(num_nodes, 1), span)


def _powerlaw_weights(num_nodes: int, avg_degree: float, max_degree: float, exponent: float,
                      device) -> torch.Tensor:
    """Expected-degree sequence w_r ~ r^(-1/(exponent-1)), clipped to max_degree, mean avg_degree."""
    r = torch.arange(1, num_nodes + 1, dtype=torch.float64, device=device)
    w = r.pow(-1.0 / (exponent - 1.0))
    w = w * (avg_degree * num_nodes / w.sum())
    for _ in range(32):  # clip + renormalise the unclipped tail to keep the mean
        over = w > max_degree
        if not bool(over.any()):
            break
        w = torch.where(over, torch.full_like(w, max_degree), w)
        free = ~over
        deficit = avg_degree * num_nodes - w.sum()
        w = torch.where(free, w * (1.0 + deficit / w[free].sum()), w)
    return w


def powerlaw_graph(num_nodes: int, num_edges: int, max_degree: int, *, exponent: float = 2.1,
                   locality: float = 0.0, window: int = 4096, seed: int = 0,
                   device="cpu", wrap: bool = True) -> CSRGraph:
    """Seeded symmetric, self-loop-free power-law (Chung-Lu) graph with ~num_edges CSR entries.

    ``num_edges // 2`` undirected pairs are drawn with endpoint probability proportional
    to a clipped power-law expected-degree sequence; node ids are randomly permuted
    ("natural" order: hubs scattered).  ``locality`` in [0, 1] redraws that fraction of
    second endpoints uniformly within ``window`` ids of the first endpoint (a
    community-ordered variant, as after Rabbit reordering); ``wrap=False`` makes the id space a
    line instead of a ring.  Pairs are symmetrised, then
    deduplicated by ``csr_from_edges`` exactly like the reference's loader.
    """
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    m = num_edges // 2
    if num_nodes == 0 or m == 0:
        e = torch.zeros(0, dtype=torch.int64, device=dev)
        return graph_from_edges(e, e, num_nodes)
    w = _powerlaw_weights(num_nodes, num_edges / num_nodes, float(max_degree), exponent, dev)
    cdf = torch.cumsum(w, 0)
    cdf = cdf / cdf[-1]
    perm = torch.randperm(num_nodes, generator=g, device=dev)
    u = torch.searchsorted(cdf, torch.rand(m, generator=g, device=dev, dtype=torch.float64)).clamp_(max=num_nodes - 1)
    v = torch.searchsorted(cdf, torch.rand(m, generator=g, device=dev, dtype=torch.float64)).clamp_(max=num_nodes - 1)
    u = perm[u]
    v = perm[v]
    if locality > 0.0:
        local = torch.rand(m, generator=g, device=dev) < locality
        off = torch.randint(-window, window + 1, (m,), generator=g, device=dev)
        near = (u + off).remainder(num_nodes) if wrap else (u + off).abs()
        if not wrap:                                   # a line of neighbourhoods instead of a ring: reflect at the ends
            near = torch.where(near >= num_nodes, 2 * (num_nodes - 1) - near, near).clamp_(0, num_nodes - 1)
        v = torch.where(local, near, v)
    keep = u != v
    u, v = u[keep], v[keep]
    src = torch.cat([u, v])
    dst = torch.cat([v, u])
    return graph_from_edges(src, dst, num_nodes)


def powerlaw_shard(num_local: int, num_global: int, num_edges: int, max_degree: int, *,
                   exponent: float = 2.1, seed: int = 0, device="cpu", locality: float = 0.0, block_start: int = 0):
    """One destination-range shard of a large power-law graph, generated independently per
    rank: ``num_local`` destination rows whose ``~num_edges`` sources are drawn from all
    ``num_global`` nodes with power-law popularity (``locality``: that fraction of the sources lies near the
    destination's own global id ``block_start + row`` instead).  Returns (row_pointers int32 [num_local+1],
    column_index int32 with GLOBAL ids), duplicates merged / columns sorted as in the loader."""
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    wr = _powerlaw_weights(num_local, num_edges / num_local, float(max_degree), exponent, dev)
    wc = _powerlaw_weights(num_global, num_edges / num_local, float(max_degree), exponent, dev)
    cdf_r = torch.cumsum(wr, 0); cdf_r = cdf_r / cdf_r[-1]
    cdf_c = torch.cumsum(wc, 0); cdf_c = cdf_c / cdf_c[-1]
    perm_r = torch.randperm(num_local, generator=g, device=dev)
    perm_c = torch.randperm(num_global, generator=g, device=dev)
    rows = perm_r[torch.searchsorted(cdf_r, torch.rand(num_edges, generator=g, device=dev, dtype=torch.float64)).clamp_(max=num_local - 1)]
    cols = perm_c[torch.searchsorted(cdf_c, torch.rand(num_edges, generator=g, device=dev, dtype=torch.float64)).clamp_(max=num_global - 1)]
    if locality > 0.0:
        # an id-local partition (what a graph partitioner / the community renumbering produces): this fraction of
        # the sources lies within +-window ids of the destination's global id instead of anywhere
        window = max(1, min(4096, num_local // 4))
        local = torch.rand(num_edges, generator=g, device=dev) < locality
        off = torch.randint(-window, window + 1, (num_edges,), generator=g, device=dev)
        near = (rows + block_start + off).clamp_(0, num_global - 1)
        cols = torch.where(local, near, cols)
    key = torch.unique(rows * num_global + cols, sorted=True)
    r = torch.div(key, num_global, rounding_mode="floor")
    c = key - r * num_global
    rp = torch.zeros(num_local + 1, dtype=torch.int64, device=dev)
    rp[1:] = torch.cumsum(torch.bincount(r, minlength=num_local), 0)
    return rp.to(torch.int32), c.to(torch.int32)


def rmat_graph(num_nodes: int, num_edges: int, *, a: float = 0.57, b: float = 0.19, c: float = 0.19, seed: int = 0,
               device="cpu", permute: bool = True) -> CSRGraph:
    """Seeded R-MAT graph (SURVEY.md 8d names it beside the Chung-Lu generator: a, b, c = 0.57, 0.19, 0.19) with
    ~num_edges CSR entries: every undirected pair picks, level by level, one quadrant of the adjacency matrix with
    probabilities (a, b, c, 1 - a - b - c); ids beyond ``num_nodes`` are folded back (modulo).  R-MAT's own numbering
    puts the hubs at the low ids and correlates the ids of neighbours; ``permute`` relabels the nodes at random (the
    "natural order" of the other generators), ``permute=False`` keeps the structured numbering.  Symmetrised and
    deduplicated by ``csr_from_edges`` like every other graph here."""
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    m = num_edges // 2
    if num_nodes == 0 or m == 0:
        e = torch.zeros(0, dtype=torch.int64, device=dev)
        return graph_from_edges(e, e, num_nodes)
    levels = max(1, (num_nodes - 1).bit_length())
    u = torch.zeros(m, dtype=torch.int64, device=dev)
    v = torch.zeros(m, dtype=torch.int64, device=dev)
    for _ in range(levels):
        r = torch.rand(m, generator=g, device=dev)
        down = r >= a + b                                   # quadrants c, d: the source's bit is 1
        right = ((r >= a) & (r < a + b)) | (r >= a + b + c)  # quadrants b, d: the destination's bit is 1
        u = u * 2 + down.to(torch.int64)
        v = v * 2 + right.to(torch.int64)
    u, v = u.remainder(num_nodes), v.remainder(num_nodes)
    if permute:
        perm = torch.randperm(num_nodes, generator=g, device=dev)
        u, v = perm[u], perm[v]
    keep = u != v
    u, v = u[keep], v[keep]
    return graph_from_edges(torch.cat([u, v]), torch.cat([v, u]), num_nodes)


def community_graph(num_nodes: int, num_edges: int, num_communities: int, *, p_in: float = 0.9, exponent: float = 2.1,
                    max_degree: int = 0, seed: int = 0, device="cpu", scramble: bool = False) -> CSRGraph:
    """Seeded community-structured graph (a degree-corrected block model): ``num_communities`` equal, contiguous id
    ranges; the first end point of a pair is drawn with power-law popularity, the second inside the first one's
    community with probability ``p_in`` (power-law inside the community too), anywhere otherwise.  ``scramble`` relabels
    the nodes at random -- the same structure with its locality hidden, which is what a renumbering has to recover."""
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    m = num_edges // 2
    if num_nodes == 0 or m == 0:
        e = torch.zeros(0, dtype=torch.int64, device=dev)
        return graph_from_edges(e, e, num_nodes)
    C = max(1, min(int(num_communities), num_nodes))
    size = (num_nodes + C - 1) // C
    w = _powerlaw_weights(num_nodes, num_edges / num_nodes, float(max_degree or num_nodes - 1), exponent, dev)
    w = w[torch.randperm(num_nodes, generator=g, device=dev)]           # popularity is independent of the community
    cdf = torch.cumsum(w, 0)
    u = torch.searchsorted(cdf, torch.rand(m, generator=g, device=dev, dtype=torch.float64) * cdf[-1]).clamp_(max=num_nodes - 1)
    v = torch.searchsorted(cdf, torch.rand(m, generator=g, device=dev, dtype=torch.float64) * cdf[-1]).clamp_(max=num_nodes - 1)
    # inside the community of u: a popularity-weighted draw from the community's own stretch of the cdf
    lo = torch.div(u, size, rounding_mode="floor") * size
    hi = (lo + size).clamp_(max=num_nodes)
    base = torch.where(lo > 0, cdf[(lo - 1).clamp_(min=0)], torch.zeros_like(cdf[lo]))
    span = cdf[hi - 1] - base
    inside = torch.searchsorted(cdf, base + torch.rand(m, generator=g, device=dev, dtype=torch.float64) * span)
    inside = torch.minimum(torch.maximum(inside, lo), hi - 1)
    v = torch.where(torch.rand(m, generator=g, device=dev) < p_in, inside, v)
    if scramble:
        perm = torch.randperm(num_nodes, generator=g, device=dev)
        u, v = perm[u], perm[v]
    keep = u != v
    u, v = u[keep], v[keep]
    return graph_from_edges(torch.cat([u, v]), torch.cat([v, u]), num_nodes)


def uniform_graph(num_nodes: int, num_edges: int, *, seed: int = 0, symmetric: bool = True,
                  device="cpu") -> CSRGraph:
    """Seeded Erdos-Renyi-style multigraph edge list (duplicates and self loops allowed in
    the raw list, merged by the CSR builder) -- used by small parity tests."""
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    m = num_edges // 2 if symmetric else num_edges
    if num_nodes == 0 or m == 0:
        e = torch.zeros(0, dtype=torch.int64, device=dev)
        return graph_from_edges(e, e, num_nodes)
    u = torch.randint(0, num_nodes, (m,), generator=g, device=dev)
    v = torch.randint(0, num_nodes, (m,), generator=g, device=dev)
    if symmetric:
        u, v = torch.cat([u, v]), torch.cat([v, u])
    return graph_from_edges(u, v, num_nodes)


# Named synthetic stand-ins for BASELINE.json's configs.  Shapes follow the
# public dataset cards; the graphs themselves are seeded power-law graphs.
CONFIGS = {
    "cora-like": dict(num_nodes=2708, num_edges=10556, max_degree=168, feat=1433, hidden=16, classes=7, seed=1),
    "citeseer-like": dict(num_nodes=3327, num_edges=9104, max_degree=99, feat=3703, hidden=16, classes=6, seed=2),
    "reddit-like": dict(num_nodes=232965, num_edges=114615892, max_degree=21657, feat=602, hidden=64, classes=41, seed=3, oversample=1.0825),
    "products-like": dict(num_nodes=2449029, num_edges=123718280, max_degree=17481, feat=100, hidden=64, classes=47, seed=4, oversample=1.0235),
    "papers100M-like": dict(num_nodes=111059956, num_edges=1615685872, max_degree=50000, feat=128, hidden=128, classes=172, seed=5, oversample=1.0),
    "amazon0505-like": dict(num_nodes=410236, num_edges=4878874, max_degree=2760, feat=96, hidden=16, classes=22, seed=6, oversample=1.019),
}


def make_config_graph(name: str, device="cpu", locality: float = 0.0, scale: float = 1.0, wrap: bool = True) -> CSRGraph:
    c = CONFIGS[name]
    n = max(2, int(c["num_nodes"] * scale))
    # symmetrisation + dedup lose a few percent of the draws on hub-hub pairs; the per-config
    # oversampling factor (calibrated on the full-size graph) brings nnz back to the dataset card's
    e = int(c["num_edges"] * scale * c.get("oversample", 1.0))
    return powerlaw_graph(n, e, min(c["max_degree"], n - 1), seed=c["seed"], locality=locality, device=device, wrap=wrap)
