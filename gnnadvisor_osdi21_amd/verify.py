"""Single-SpMM verification and profiling harness (``Verification``).

API counterpart of the reference's GNNAdvisor/unitest.py:9-79: constructor arguments,
``reference / compute / compare / profile_spmm`` and the printed lines
(``# Verification PASSED|FAILED``, ``=> SpMM profiling avg (ms): ...``) are kept so that
log scrapers keep working.  The pass criterion is the reference's: on X = ones, fewer
than 1e-4 of the elements may differ *exactly* from the CPU edge-list sum (:54-63).

The reference's CPU side is ``torch_sparse.spmm`` (an unpinned pip dependency that is
not installed here); what it computes at that call site -- out[row] += val * X[col] over
the raw edge list, duplicates included -- is written out in ``reference``.
"""
import time

import torch

from . import load_extension

GNNA = load_extension()

WARMUP_CALLS = 10   # unitest.py:69


class Verification(object):
    def __init__(self, dim, row_pointers, column_index, degrees, partPtr, part2Node,
                 partSize, dimWorker, warpPerBlock):
        self.graph = (row_pointers, column_index, degrees, partPtr, part2Node)
        self.knobs = (partSize, dimWorker, warpPerBlock)
        self.row_pointers, self.column_index, self.degrees = row_pointers, column_index, degrees
        self.partPtr, self.part2Node = partPtr, part2Node
        self.partSize, self.dimWorker, self.warpPerBlock = partSize, dimWorker, warpPerBlock
        self.num_nodes = len(row_pointers) - 1
        self.test_embedding = self.output_embedding = dim
        self.X = torch.ones(self.num_nodes, dim)
        self.W = torch.ones(dim, dim)
        self.result = None
        self.result_ref = None

    def _sag(self, X_dev):
        return GNNA.SAG(X_dev, *self.graph, *self.knobs)

    def reference(self, column_index, val, num_nodes):
        """CPU edge-list sum: ``column_index`` is the raw [2, E] (src, dst) list, ``val`` its
        E values (all ones in the reference's driver)."""
        print("# Compute reference on CPU")
        edges = torch.as_tensor(column_index, dtype=torch.int64)
        weights = torch.as_tensor(val, dtype=torch.float32).reshape(-1, 1)
        acc = torch.zeros(num_nodes, self.X.size(1))
        acc.index_add_(0, edges[0], weights * self.X.index_select(0, edges[1]))
        self.result_ref = acc

    def compute(self):
        print("# Compute result on GPU")
        self.result = self._sag(self.X.cuda())

    def compare(self):
        if self.result_ref is None or self.result is None:
            raise ValueError("MUST compute result and result reference (CPU) first!!")
        mismatch = (self.result_ref != self.result.cpu()).float().mean().item()
        ok = mismatch < 1e-4
        print("# Verification PASSED" if ok else "# Verification FAILED")
        return ok

    def profile_spmm(self, round=1):
        """Wall-clock ms per SAG call including allocation and binding overhead, exactly what
        the reference's number contains (unitest.py:65-79)."""
        X = self.X.cuda()
        print("SpMM profiling size: N: {}, N: {}, K: {}".format(X.size(0), X.size(0), X.size(1)))
        for _ in range(WARMUP_CALLS):
            self.result = self._sag(X)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(round):
            self.result = self._sag(X)
        torch.cuda.synchronize()
        avg_ms = (time.perf_counter() - t0) * 1e3 / round
        print("=> SpMM profiling avg (ms): {:.3f}".format(avg_ms))
        print()
        return avg_ms
