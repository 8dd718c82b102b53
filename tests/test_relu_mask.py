"""The "two-rank wrong-gradient transient" of rounds 3-4, reproduced on purpose (VERDICT r4 task 1).

What was seen: in ~1.5 % of the two-process sessions one rank's INPUT gradient of the layer step
`GIN(relu(GCN(F)))` had 3-15 rows off by 2.6e-4 ... 5.9e-2 of the bound's scale, while the layer outputs, both
weight gradients and every standalone aggregation were inside 1e-4.  Cause: not the kernels, not the exchange, not the
BLAS library -- the CHECKER.  `gcn_gin_reference` took relu' from the sign of the fp64 pre-activation H1 = Ahat (F W1);
an element of H1 that cancels to within fp32 rounding of zero has no defined sign in fp32 (it depends on the summation
order, i.e. on the order the kernels' float atomics land in), and ONE disagreeing element of the mask moves dF in every
neighbour row of that node by far more than 1e-4 of the scale, while out (relu is continuous), dW2 (no mask) and dW1
(a sum over all rows) stay inside the bound -- exactly the observed picture.

Here the cancellation is constructed (W1 columns in the null space of a few rows of Ahat F), so it happens every time:
* CPU: one flipped mask element reproduces "a few rows of dF off, everything else fine" in the fp64 formulas;
* GPU: the real kernels, 32 cancelling elements -- the strict fp64 mask mismatches in neighbour rows only, the reference
  that takes the computed sign inside the bound (what the layer tests use now) agrees."""
import numpy as np
import pytest
import torch

from gnnadvisor_osdi21_amd import graph
from util import gcn_gin_reference


def _cancelling_weights(g, F, hidden, per_column, seed):
    """W1 [fin, hidden] (fp32) such that H1[i, k] = (Ahat F W1)[i, k] = 0 in exact arithmetic for `per_column` chosen
    low-degree nodes i of every column k.  -> (W1, [(i, k)])"""
    import scipy.sparse as sp
    n, fin = F.shape
    rp, ci, deg = g.row_pointers.numpy(), g.column_index.numpy(), g.degrees.double().numpy()
    A = sp.csr_matrix((np.ones(len(ci)), ci, rp), shape=(n, n))
    AF = sp.diags(deg) @ A @ sp.diags(deg) @ F.double().numpy()
    rng = np.random.default_rng(seed)
    rowdeg = rp[1:] - rp[:-1]
    cand = np.nonzero((rowdeg >= 6) & (rowdeg <= 14))[0]
    W1 = rng.uniform(-1, 1, (fin, hidden)) / np.sqrt(hidden)
    picked = []
    for k in range(hidden):
        rows = rng.choice(cand, per_column, replace=False)
        C = AF[rows]                                            # per_column x fin constraints on column k
        w = W1[:, k]
        W1[:, k] = w - np.linalg.pinv(C) @ (C @ w)              # projection onto the null space of C
        picked += [(int(i), k) for i in rows]
    return torch.from_numpy(W1).float(), picked


def test_one_flipped_mask_element_moves_a_few_rows_of_dF_only():
    g = graph.powerlaw_graph(4000, 40000, 300, seed=5)
    F = torch.randn(g.num_nodes, 12, generator=torch.Generator().manual_seed(3))
    W1, picked = _cancelling_weights(g, F, 8, 1, seed=1)
    W2 = torch.randn(8, 5, generator=torch.Generator().manual_seed(4)) * 0.4
    wgt = torch.linspace(0.5, 1.5, 5)
    base = gcn_gin_reference(g, F, W1, W2, wgt)
    assert base["ambiguous"] >= len(picked) and base["min_ratio"] < 1e-6      # the constructed elements cancel
    H1 = torch.from_numpy(base["H1"][0]).clone()
    i, k = picked[0]
    H1[i, k] = -H1[i, k] if H1[i, k] != 0 else 1.0                           # "the fp32 path computed the other sign"
    flipped = gcn_gin_reference(g, F, W1, W2, wgt, H1_got=H1)
    assert flipped["sign_flips"] == 1
    rel = {name: np.abs(flipped[name][0] - base[name][0]) / np.maximum(1.0, base[name][1]) for name in ("out", "dF", "dW1", "dW2")}
    rows_off = np.nonzero(rel["dF"].max(axis=1) > 1e-4)[0]
    rp, ci = g.row_pointers.numpy(), g.column_index.numpy()
    neighbours = set(ci[rp[i]:rp[i + 1]].tolist())
    assert 1 <= len(rows_off) <= len(neighbours) and set(rows_off.tolist()) <= neighbours      # a few rows: neighbours of node i
    assert rel["out"].max() == 0 and rel["dW2"].max() == 0 and rel["dW1"].max() < 1e-4            # everything else inside the bound


@pytest.mark.gpu
def test_cancelling_preactivations_on_the_real_kernels():
    from gnnadvisor_osdi21_amd import ops
    from test_module_gpu import _info
    g = graph.powerlaw_graph(4000, 40000, 300, seed=5)
    fin, hid, ncls = 12, 8, 5
    F = torch.randn(g.num_nodes, fin, generator=torch.Generator().manual_seed(3))
    W1, picked = _cancelling_weights(g, F, hid, 4, seed=2)
    info, _, _ = _info(g, fin, hid)
    c1, c2 = ops.GCNConv(fin, hid).cuda(), ops.GINConv(hid, ncls).cuda()
    with torch.no_grad():
        c1.weights.copy_(W1)
    wgt = torch.linspace(0.5, 1.5, ncls, device="cuda")
    Fd = F.cuda().requires_grad_(True)
    h1 = c1(Fd, info)
    y = c2(torch.relu(h1), info)
    (y * wgt).sum().backward()
    strict = gcn_gin_reference(g, F, c1.weights, c2.weights, wgt)
    aware = gcn_gin_reference(g, F, c1.weights, c2.weights, wgt, H1_got=h1)
    assert len(picked) <= aware["ambiguous"] <= len(picked) + 16
    got = dict(out=y, H1=h1, dF=Fd.grad, dW1=c1.weights.grad, dW2=c2.weights.grad)

    def rel(ref, name):
        return np.abs(got[name].detach().double().cpu().numpy() - ref[name][0]) / np.maximum(1.0, ref[name][1])
    for name in got:                                            # with the computed sign inside the bound: all green
        assert rel(aware, name).max() <= 1e-4, (name, rel(aware, name).max())
    # and the strict fp64 mask shows the transient's signature: of 32 elements at fp32's rounding level some come out
    # with the other sign (p(none) ~ 1e-5), dF is then off in neighbour rows of those nodes and nowhere else
    assert aware["sign_flips"] >= 1
    H1_ref, H1_abs = strict["H1"]
    flipped = np.argwhere((h1.detach().cpu().numpy() > 0) != (H1_ref > 0))
    assert len(flipped) == aware["sign_flips"]                  # every disagreement lies inside the band
    assert (np.abs(H1_ref[flipped[:, 0], flipped[:, 1]]) <= 1e-5 * H1_abs[flipped[:, 0], flipped[:, 1]]).all()
    off = np.nonzero(rel(strict, "dF").max(axis=1) > 1e-4)[0]
    rp, ci = g.row_pointers.numpy(), g.column_index.numpy()
    allowed = set()
    for i in set(flipped[:, 0].tolist()):
        allowed |= set(ci[rp[i]:rp[i + 1]].tolist())
    assert len(off) >= 1 and set(off.tolist()) <= allowed
    for name in ("out", "H1", "dW2"):
        assert rel(strict, name).max() <= 1e-4
