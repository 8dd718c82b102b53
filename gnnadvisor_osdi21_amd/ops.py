"""Operator layer over the ``GNNAdvisor`` extension: autograd functions and the GCN / GIN
convolution modules.

API counterpart of the reference's GNNAdvisor/gnn_conv.py -- the names and call
signatures its driver uses are kept (``ScatterAndGather`` :7-27, ``GNNAFunction`` :30-78,
``GCNConv`` :80-98, ``GNNAFunction_GIN`` :101-126, ``GINConv`` :128-147; weights drawn from
U(-1/sqrt(out), 1/sqrt(out)) :86-88; GIN epsilon fixed at 0.5 and spelled ``eplison``
:132) -- but the implementation is this package's own: every op unpacks the graph
bundle once through ``_graph_args`` and calls the HIP extension.  No fallback exists.
"""
import math

import torch
from torch.autograd import Function
from torch.nn import Module, Parameter

from . import load_extension

GNNA = load_extension()


def _graph_args(info):
    """(row_pointers, column_index, degrees, partPtr, part2Node) of a decider.inputProperty."""
    return (info.row_pointers, info.column_index, info.degrees, info.partPtr, info.part2Node)


def _knobs(info):
    return (info.partSize, info.dimWorker, info.warpPerBlock)


class ScatterAndGather(Function):
    """Y = A X (unweighted neighbor sum).  A is assumed symmetric, so backward is the same op."""

    @staticmethod
    def forward(ctx, X, inputInfo):
        ctx.graph, ctx.knobs = _graph_args(inputInfo), _knobs(inputInfo)
        return GNNA.SAG(X, *ctx.graph, *ctx.knobs)

    @staticmethod
    def backward(ctx, d_output):
        return GNNA.SAG(d_output.contiguous(), *ctx.graph, *ctx.knobs), None


class GNNAFunction(Function):
    """GCN layer: dense update X W, then degree-weighted aggregation (update -> aggregate)."""

    @staticmethod
    def forward(ctx, X, weight, inputInfo):
        ctx.save_for_backward(X, weight)
        ctx.graph, ctx.knobs = _graph_args(inputInfo), _knobs(inputInfo)
        return GNNA.forward(X, weight, *ctx.graph, *ctx.knobs)[0]

    @staticmethod
    def backward(ctx, d_output):
        X, weight = ctx.saved_tensors
        return _gcn_backward(ctx, d_output.contiguous(), X, weight)


def _gcn_backward(ctx, d_output, X, weight):
    """(d_input, d_weight, None) of a GCN layer from the gradient of its (pre-activation) output."""
    if not ctx.needs_input_grad[0]:
        # first layer: the features need no gradient (the reference computes and drops d_input)
        return None, GNNA.backward_weight(d_output, X, *ctx.graph, *ctx.knobs)[0], None
    d_input, d_weight = GNNA.backward(d_output, X, weight, *ctx.graph, *ctx.knobs)
    return d_input, d_weight, None


class GNNAFunction_ReLU(Function):
    """relu(GCN layer) in one pass: the aggregation's epilogue clamps every row where it is written (libgnna
    GNNA_EPILOGUE_RELU; reference call site F.relu(conv1(...)), GNNA_main.py:151) -- one elementwise pass over [N, hidden]
    less per layer.  Backward masks dY with (Y > 0), which the saved output itself provides."""

    @staticmethod
    def forward(ctx, X, weight, inputInfo):
        rp, ci, deg, pp, p2n = _graph_args(inputInfo)
        ctx.graph, ctx.knobs = (rp, ci, deg, pp, p2n), _knobs(inputInfo)
        Y = GNNA.aggregate_ld(1, torch.mm(X, weight), ci, deg, 1.0, pp, p2n, inputInfo.partSize, None, False, True)
        ctx.save_for_backward(X, weight, Y)
        return Y

    @staticmethod
    def backward(ctx, d_output):
        X, weight, Y = ctx.saved_tensors
        return _gcn_backward(ctx, d_output * (Y > 0), X, weight)


class GNNAFunction_GIN(Function):
    """GIN layer: epsilon-scaled aggregation T = eps A X, then update T W (aggregate -> update).
    T is what backward needs, so it is saved instead of X (reference gnn_conv.py:109,119)."""

    @staticmethod
    def forward(ctx, X, weight, inputInfo, eplison):
        rp, ci, _deg, pp, p2n = _graph_args(inputInfo)
        ctx.graph, ctx.knobs, ctx.eplison = (rp, ci, pp, p2n), _knobs(inputInfo), eplison
        X_prime, X_agg = GNNA.forward_gin(X, weight, rp, ci, eplison, pp, p2n, *ctx.knobs)
        ctx.save_for_backward(X_agg, weight)
        return X_prime

    @staticmethod
    def backward(ctx, d_output):
        X_agg, weight = ctx.saved_tensors
        rp, ci, pp, p2n = ctx.graph
        if not ctx.needs_input_grad[0]:
            # first layer: d_weight = T^T dY needs no aggregation at all; the reference aggregates
            # dY W^T at the input width (F = 602 on Reddit) and drops the result
            return None, GNNA.xtg(X_agg, d_output.contiguous()), None, None
        d_input, d_weight = GNNA.backward_gin(d_output.contiguous(), X_agg, weight, rp, ci,
                                              ctx.eplison, pp, p2n, *ctx.knobs)
        return d_input, d_weight, None, None


def _mm_for_gather(X, weight, column_index):
    """X W written in the layout the following (unweighted) aggregation gathers best: rows with the leading dimension
    `gnna_preferred_ld` names (e.g. 128 floats for 64-float rows that are gathered hundreds of times -- every row on its own
    512-byte boundary), so that the library needs no staged copy of them.  rocBLAS writes a leading dimension for free."""
    from . import _lib
    n, dim = X.shape[0], weight.shape[1]
    ld = _lib.preferred_ld(dim, n, column_index.numel())
    if ld == dim:
        return torch.mm(X, weight)
    out = _lib.empty_rows(n, dim, ld, X.device)
    torch.mm(X, weight, out=out)
    return out


def _row_units(width: int) -> int:
    """Cost of aggregating one neighbor row of `width` floats, in 64-float wavefront sweeps."""
    return (int(width) + 63) // 64


class GNNAFunction_GIN_UpdateFirst(Function):
    """The same GIN layer, Y = (eps A X) W, evaluated as eps A (X W): identical function (the
    layer's "MLP" is one linear map, so it commutes with the neighbor sum; only the fp32
    association differs), but both aggregations -- forward on X W, backward on dY -- run at the
    output width.  Chosen by GINConv when the layer narrows (Reddit layer 1: 602 -> 64)."""

    @staticmethod
    def forward(ctx, X, weight, inputInfo, eplison, relu=False):
        rp, ci, _deg, pp, p2n = _graph_args(inputInfo)
        ctx.graph, ctx.knobs, ctx.eplison = (rp, ci, pp, p2n), _knobs(inputInfo), eplison
        ctx.relu = bool(relu)
        with torch.no_grad():
            XW = _mm_for_gather(X, weight, ci)
        Y = GNNA.aggregate_ld(2, XW, ci, None, eplison, pp, p2n, inputInfo.partSize, None, False, bool(relu))
        if relu:      # (the aggregation is the layer's last step here, so its epilogue can clamp)
            ctx.save_for_backward(X, weight, Y)
        else:
            ctx.save_for_backward(X, weight)
        return Y

    @staticmethod
    def backward(ctx, d_output):
        X, weight = ctx.saved_tensors[:2]
        rp, ci, pp, p2n = ctx.graph
        if ctx.relu:
            d_output = d_output * (ctx.saved_tensors[2] > 0)
        G = GNNA.aggregate_gin(d_output.contiguous(), rp, ci, ctx.eplison, pp, p2n, *ctx.knobs)   # A symmetric
        d_input = torch.mm(G, weight.t()) if ctx.needs_input_grad[0] else None
        return d_input, GNNA.xtg(X, G), None, None, None


class _NeighborConv(Module):
    """Shared parameter handling of the two convolution modules."""

    def __init__(self, input_dim, output_dim):
        super().__init__()
        self.weights = Parameter(torch.empty(input_dim, output_dim))
        self.reset_parameters()

    def reset_parameters(self):
        bound = 1.0 / math.sqrt(self.weights.size(1))
        with torch.no_grad():
            self.weights.uniform_(-bound, bound)


class GCNConv(_NeighborConv):
    def forward(self, X, inputInfo, relu=False):
        """X: [num_nodes, input_dim]; inputInfo: decider.inputProperty holding the CSR, the
        sqrt-degree vector and the neighbor-group partition on X's device.  relu=True returns relu(layer) with the
        clamp fused into the aggregation (same values as F.relu(conv(X, inputInfo)))."""
        return (GNNAFunction_ReLU if relu else GNNAFunction).apply(X, self.weights, inputInfo)


class GINConv(_NeighborConv):
    def __init__(self, input_dim, output_dim, update_first="auto"):
        """update_first: False = the reference's order (aggregate at the input width, then X W);
        True = update first; "auto" = whichever aggregates fewer 64-float sweeps per neighbor."""
        self.eplison = 0.5
        self.update_first = update_first
        super().__init__(input_dim, output_dim)

    def _use_update_first(self, X) -> bool:
        if self.update_first != "auto":
            return bool(self.update_first)
        fin, fout = _row_units(self.weights.size(0)), _row_units(self.weights.size(1))
        # aggregate-first: forward at Fin, plus backward at Fin when the input needs a gradient;
        # update-first: forward and backward at Fout
        needs_dx = X.requires_grad and torch.is_grad_enabled()
        return 2 * fout < (2 * fin if needs_dx else fin)

    def forward(self, X, inputInfo, relu=False):
        """relu=True returns relu(layer): fused into the aggregation when the layer runs update-first (the aggregation is
        its last step), an ordinary F.relu behind the dense update otherwise."""
        if self._use_update_first(X):
            return GNNAFunction_GIN_UpdateFirst.apply(X, self.weights, inputInfo, self.eplison, relu)
        Y = GNNAFunction_GIN.apply(X, self.weights, inputInfo, self.eplison)
        return torch.relu(Y) if relu else Y
