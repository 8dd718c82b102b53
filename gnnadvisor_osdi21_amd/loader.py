"""Graph loading: ``custom_dataset`` (counterpart of the reference's GNNAdvisor/dataset.py:20-176).

Same constructor (``path, dim, num_class, load_from_txt, verbose``), same public fields read
by the driver and the Decider (``num_nodes, num_edges, num_features, num_classes, edge_index,
avg_degree, avg_edgeSpan, row_pointers, column_index, degrees, x, y, val, reorder_flag``) and the
``rabbit_reorder()`` hook.  Formats (dataset.py:59-91): a text edge list with ``src dst`` per
line, or an ``.npz`` with keys ``src_li``, ``dst_li``, ``num_nodes``.

What is this package's own: CSR, degrees and edge statistics come from the native host
builders of libgnna (counting sort + per-row sort/unique instead of scipy; ``sqrt(max(deg,1))``),
renumbering is the native community renumbering ``gnna_reorder_community_i32`` (label propagation +
community chain + barycentre sweeps, multi-threaded; ``reorder_method = "rcm"`` selects the reverse
Cuthill-McKee sweep instead) in place of the third-party Rabbit Order module (same contract: a relabelled
``[2, E]`` int32 edge list whose CSR and degrees are then rebuilt, dataset.py:147-172), and tensors go to ``device`` instead of an
unconditional ``.cuda()``.  ``from_edges`` / ``from_synthetic`` build datasets without files (no
dataset ships with the reference and there is no network here).
"""
from __future__ import annotations

import time

import numpy as np
import torch

from . import _lib


def _default_device():
    return torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")


class custom_dataset(torch.nn.Module):
    def __init__(self, path, dim, num_class, load_from_txt=True, verbose=False, device=None,
                 _edges=None):
        super().__init__()
        self.load_from_txt = load_from_txt
        self.num_nodes = 0
        self.num_features = dim
        self.num_classes = num_class
        self.edge_index = None
        self.reorder_flag = False
        self.reorder_method = "community"                      # or "rcm" (reverse Cuthill-McKee)
        # True: rabbit_reorder() also moves the per-node data (x, y, the masks) to the new ids, so that row i of x is
        # still node i's embedding.  The reference leaves them where they were (dataset.py:138-172 relabels the edge
        # list only), which goes unnoticed there because x is random and y constant; the mi355x Decider sets this.
        self.permute_node_data = False
        self.new_id = None
        self.verbose_flag = verbose
        self.avg_degree = -1
        self.avg_edgeSpan = -1
        self.device = torch.device(device) if device is not None else _default_device()

        if _edges is not None:
            src, dst, n = _edges
            self._set_edges(np.asarray(src), np.asarray(dst), int(n))
        else:
            self.init_edges(path)
        self.init_embedding(dim)
        self.init_labels(num_class)

        n = self.num_nodes
        mask = lambda frac: torch.arange(n, device=self.device) < int(n * frac)   # dataset.py:45-53
        self.train_mask, self.val_mask, self.test_mask = mask(1), mask(0.3), mask(0.1)

    # ------------------------------------------------------------------ construction helpers
    @classmethod
    def from_edges(cls, src, dst, num_nodes, dim, num_class, verbose=False, device=None):
        return cls(None, dim, num_class, load_from_txt=False, verbose=verbose, device=device,
                   _edges=(src, dst, num_nodes))

    @classmethod
    def from_synthetic(cls, name, dim=None, num_class=None, scale=1.0, verbose=False, device=None, locality=0.0, scramble=False):
        """One of graph.CONFIGS (seeded power-law stand-ins for the BASELINE.json graphs).  locality: that share of the edges
        stays within +-4096 ids of a hidden order; scramble: the ids are then relabelled at random -- a graph with community
        structure whose numbering hides it, which is what the renumbering has to recover."""
        from . import graph
        c = graph.CONFIGS[name]
        g = graph.make_config_graph(name, device="cuda" if torch.cuda.is_available() else "cpu", scale=scale, locality=locality)
        rows = torch.repeat_interleave(torch.arange(g.num_nodes, device=g.row_pointers.device, dtype=torch.int32),
                                       (g.row_pointers[1:] - g.row_pointers[:-1]).long())
        cols = g.column_index
        if scramble:
            perm = torch.randperm(g.num_nodes, device=rows.device, generator=torch.Generator(device=rows.device).manual_seed(1)).to(torch.int32)
            rows, cols = perm[rows.long()], perm[cols.long()]
        return cls.from_edges(rows.cpu().numpy(), cols.cpu().numpy(), g.num_nodes,
                              dim if dim is not None else c["feat"],
                              num_class if num_class is not None else c["classes"], verbose, device)

    # ------------------------------------------------------------------ loading (dataset.py:55-122)
    def init_edges(self, path):
        start = time.perf_counter()
        if self.load_from_txt:
            pairs = np.loadtxt(path, dtype=np.int64, ndmin=2, usecols=(0, 1))
            src, dst = pairs[:, 0], pairs[:, 1]
            n = int(max(src.max(), dst.max())) + 1 if len(src) else 0
            tag = "txt"
        else:
            if not str(path).endswith(".npz"):
                raise ValueError("graph file must be a .npz file")
            obj = np.load(path)
            src, dst, n = obj["src_li"], obj["dst_li"], int(obj["num_nodes"])
            tag = "npz"
        if self.verbose_flag:
            print("# Loading ({})(s): {:.3f}".format(tag, time.perf_counter() - start))
        self._set_edges(src, dst, n)

    def _set_edges(self, src, dst, num_nodes):
        self.num_nodes = int(num_nodes)
        self.num_edges = int(len(src))
        # [2, E] like the reference's (dataset.py:96), but int32 (the CSR and the kernels are int32: ids that do not fit are refused
        # here instead of wrapping there; at 1.1e8 edges every int64 copy / conversion of this array is a second of a host thread)
        if self.num_edges and (self.num_nodes > 2**31 - 1 or int(np.min(src)) < 0 or int(np.min(dst)) < 0
                               or max(int(np.max(src)), int(np.max(dst))) >= self.num_nodes):
            raise ValueError("node ids must lie in [0, num_nodes) with num_nodes < 2^31")
        self.edge_index = np.empty((2, self.num_edges), dtype=np.int32)
        self.edge_index[0], self.edge_index[1] = src, dst
        self.avg_degree = self.num_edges / self.num_nodes if self.num_nodes else 0.0
        self.avg_edgeSpan = _lib.edge_span(self.edge_index[0], self.edge_index[1])
        if self.verbose_flag:
            print('# nodes: {}'.format(self.num_nodes))
            print("# avg_degree: {:.2f}".format(self.avg_degree))
            print("# avg_edgeSpan: {}".format(int(self.avg_edgeSpan)))
        self.val = np.ones(self.num_edges, dtype=np.float32)   # (dataset.py:106 builds [1] * E: the same values, not 1e8 Python ints)
        self._build_csr("# Build CSR (s): {:.3f}")

    def _build_csr(self, msg):
        start = time.perf_counter()
        self.row_pointers, self.column_index = _lib.csr_from_edges(self.edge_index[0], self.edge_index[1],
                                                                   self.num_nodes)
        if self.verbose_flag:
            print(msg.format(time.perf_counter() - start))
        self.degrees = _lib.degrees(self.row_pointers).to(self.device)

    def init_embedding(self, dim):
        """Random node embeddings (dataset.py:124-129)."""
        self.x = torch.randn(self.num_nodes, dim, device=self.device)

    def init_labels(self, num_class):
        """All-ones labels (dataset.py:131-136)."""
        self.y = torch.ones(self.num_nodes, dtype=torch.long, device=self.device)

    # ------------------------------------------------------------------ renumbering hook (dataset.py:138-172)
    def rabbit_reorder(self):
        """If the Decider set ``reorder_flag``: renumber nodes for locality, then rebuild CSR
        and degrees; otherwise do nothing."""
        if not self.reorder_flag:
            if self.verbose_flag:
                print("Reorder flag is not set. Skipped...")
            return
        start = time.perf_counter()
        if self.reorder_method == "rcm":
            new_id = _lib.reorder_rcm(self.edge_index[0], self.edge_index[1], self.num_nodes)
        else:
            # from the CSR this dataset already holds (symmetric graphs: it IS the algorithm's adjacency; same permutation as
            # from the edge list, gnna_reorder_community_csr_i32)
            new_id = _lib.reorder_community_csr(self.row_pointers, self.column_index, self.num_nodes)
        self.new_id = new_id.numpy().astype(np.int64)              # new_id[old] (kept for callers that hold per-node data)
        if self.verbose_flag:
            print("# Reorder time (s): {}".format(time.perf_counter() - start))
        t1 = time.perf_counter()
        # edge list relabelled in place (+ its new span), CSR rows permuted / mapped / re-sorted: no global sort again
        self.avg_edgeSpan_after = _lib.relabel_edges_(torch.from_numpy(self.edge_index[0]), torch.from_numpy(self.edge_index[1]),
                                                      new_id, self.num_nodes)
        self.row_pointers, self.column_index = _lib.relabel_csr(self.row_pointers, self.column_index, new_id, self.num_nodes)
        self.degrees = _lib.degrees(self.row_pointers).to(self.device)
        if self.verbose_flag:
            print("# Re-Build CSR (s): {:.3f}".format(time.perf_counter() - t1))
        new_id = self.new_id
        if self.permute_node_data:
            where = torch.from_numpy(new_id).to(self.device)   # new position of old row i
            for name in ("x", "y", "train_mask", "val_mask", "test_mask"):
                old = getattr(self, name, None)
                if old is not None and old.shape[0] == self.num_nodes:
                    moved = torch.empty_like(old)
                    moved[where] = old
                    setattr(self, name, moved)
        self.reorder_seconds = time.perf_counter() - start


# ---------------------------------------------------------------------------------------------- sharded ingestion
def read_edge_list(path, load_from_txt: bool):
    """The reference's two file formats (dataset.py:59-91): -> (src, dst, num_nodes) as numpy arrays."""
    if load_from_txt:
        pairs = np.loadtxt(path, dtype=np.int64, ndmin=2, usecols=(0, 1))
        src, dst = pairs[:, 0], pairs[:, 1]
        return src, dst, (int(max(src.max(), dst.max())) + 1 if len(src) else 0)
    if not str(path).endswith(".npz"):
        raise ValueError("graph file must be a .npz file")
    obj = np.load(path)
    return obj["src_li"], obj["dst_li"], int(obj["num_nodes"])


class GraphShard:
    """One rank's part of a graph file: the CSR rows of its destination range (local int32 row_pointers, GLOBAL
    int32 column ids), the nnz-balanced row bounds of all ranks, and the whole-graph statistics the Decider reads."""

    def __init__(self, num_nodes, bounds, rank, row_pointers, column_index, num_edges, avg_degree, avg_edgeSpan, new_id):
        self.num_nodes, self.bounds, self.rank = int(num_nodes), [int(b) for b in bounds], int(rank)
        self.row_pointers, self.column_index = row_pointers, column_index
        self.num_edges, self.avg_degree, self.avg_edgeSpan = int(num_edges), float(avg_degree), float(avg_edgeSpan)
        self.new_id = new_id                                   # new_id[old id] when the nodes were renumbered, else None
        self.degrees = _lib.degrees(row_pointers)              # sqrt(max(deg, 1)) of the local rows (dataset.py:121-122)

    @property
    def row_range(self):
        return self.bounds[self.rank], self.bounds[self.rank + 1]


def load_graph_shard(path, rank: int, world: int, load_from_txt: bool = False, reorder: bool = False,
                     reorder_method: str = "community", share_fn=None, verbose: bool = False,
                     _edges=None) -> GraphShard:
    """Sharded counterpart of ``custom_dataset`` (dataset.py:55-122) for one-process-per-GPU jobs: every rank reads the
    edge list, the rows are cut into ``world`` contiguous nnz-balanced blocks from 64-bit row counts (an edge list of
    more than 2^31 entries -- papers100M symmetrised -- is fine; a single shard must stay below 2^31), and this rank
    builds ONLY the CSR of its own block (duplicates merged, columns sorted: the loader's semantics).

    ``reorder``: the nodes are renumbered for locality first (``gnna_reorder_community_i32``; the role of
    dataset.py:138-172) -- by ONE rank, the others receive ``new_id`` through ``share_fn(tensor_or_None, src_rank=0)``
    (e.g. a ``torch.distributed`` broadcast), so the permutation is computed once and is the same everywhere.  Fewer
    remote sources per shard is what the halo exchange lives on."""
    t0 = time.perf_counter()
    src, dst, n = _edges if _edges is not None else read_edge_list(path, load_from_txt)
    # node ids are int32 in the CSR and in the kernels: refuse ids that the cast below would wrap silently
    top = int(max(np.max(src), np.max(dst))) if len(src) else -1
    low = int(min(np.min(src), np.min(dst))) if len(src) else 0
    if int(n) > 2**31 - 1 or top >= 2**31 - 1 or low < 0 or top >= int(n):
        raise ValueError(f"node ids must lie in [0, num_nodes) with num_nodes < 2^31 (got ids {low}..{top}, num_nodes {n})")
    src = np.ascontiguousarray(src, dtype=np.int32)
    dst = np.ascontiguousarray(dst, dtype=np.int32)
    num_edges = int(len(src))
    new_id = None
    if reorder:
        if share_fn is None or rank == 0:
            renumber = _lib.reorder_rcm if reorder_method == "rcm" else _lib.reorder_community
            new_id = renumber(src, dst, n)
        if share_fn is not None:
            new_id = share_fn(new_id if rank == 0 else torch.empty(n, dtype=torch.int32), 0)
        perm = new_id.numpy()
        src, dst = perm[src], perm[dst]
    counts = _lib.row_counts(src, n)
    bounds = _lib.row_splits(counts, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    rp, ci = _lib.csr_from_edges_range(src, dst, n, lo, hi, capacity=int(counts[lo:hi].sum()))
    span = _lib.edge_span(src, dst)
    if verbose:
        print("# rank {}: rows [{}, {}) of {}, {} of {} edges, {:.3f} s".format(rank, lo, hi, n, ci.numel(), num_edges,
                                                                                 time.perf_counter() - t0))
    return GraphShard(n, bounds, rank, rp, ci, num_edges, num_edges / n if n else 0.0, span, new_id)
