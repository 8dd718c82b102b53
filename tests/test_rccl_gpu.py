"""The RCCL path on the one GPU of the test box: a ONE-rank `nccl` process group (created with `device_id`) through
which `ShardedAggregator(force_collectives=True)` still issues every collective of the N-rank step --
`all_gather_into_tensor` (whole, and K pieces into adjacent views of one buffer), `all_to_all_single` with split
lists, the all-reduces of the set-up decisions and of the weight gradient.  The second half of the rank's own block
is the "remote" part: it reaches the kernels only through the collective, so a missing stream dependency between
RCCL's stream and the library's launches shows as NaN / stale rows (the receive buffers are poisoned before every
step).  Results are compared with the oracle (checker only).  Reference: none -- the reference is single-GPU
(GNNA_main.py:53); SURVEY 8(e)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


_PROBE = r"""
import os, sys, torch, torch.distributed as dist
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d", rank=0, world_size=1, device_id=dev)
t = torch.ones(1024, device=dev)
dist.all_reduce(t)
torch.cuda.synchronize()
assert float(t.sum()) == 1024.0
dist.destroy_process_group()
print("RCCL-ONE-RANK-OK")
"""


@pytest.fixture(scope="module")
def rccl_group():
    """One-rank nccl (= RCCL) process group in THIS process.  A separate process tries it first under a time limit, so
    that a hang inside the library's bootstrap fails this module instead of stalling the whole run."""
    assert os.environ.get("HIP_LAUNCH_BLOCKING", "0") in ("", "0"), "these tests are about asynchronous ordering"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        r = subprocess.run([sys.executable, "-c", _PROBE % _free_port()], cwd=ROOT, env=env, capture_output=True,
                           text=True, timeout=240)
    except subprocess.TimeoutExpired:
        pytest.fail("a one-rank nccl process group did not come up within 240 s")
    assert r.returncode == 0 and "RCCL-ONE-RANK-OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert not dist.is_initialized()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1, device_id=dev)
    try:
        yield dev
    finally:
        torch.cuda.synchronize()
        dist.destroy_process_group()


def _graph_and_refs(n=20000, e=1600000, D=64, seed=41):
    import oracle
    from gnnadvisor_osdi21_amd import graph
    g = graph.powerlaw_graph(n, e, 3000, seed=seed)
    X = torch.randn(n, D, generator=torch.Generator().manual_seed(seed + 1))
    rpn, cin, degn, Xn = g.row_pointers.numpy(), g.column_index.numpy(), g.degrees.numpy(), X.numpy()
    refs = {}
    for mode, eps in ((0, 1.0), (1, 1.0), (2, 0.5)):
        refs[mode] = (oracle.csr_f64(mode, Xn, rpn, cin, degn, eps),
                      np.maximum(1.0, oracle.csr_f64(mode, np.abs(Xn), rpn, cin, degn, eps)), eps)
    return g, X, refs


def _poison_receive_buffers(agg):
    for b in (agg._gather_buf, agg._halo_buf):
        if b is not None:
            b.fill_(float("nan"))


@pytest.mark.parametrize("chunks,exchange", [(1, "allgather"), (3, "allgather"), (1, "halo"), (2, "halo")])
def test_every_exchange_form_through_a_one_rank_rccl_group(rccl_group, chunks, exchange):
    from gnnadvisor_osdi21_amd.dist import ShardedAggregator
    dev = rccl_group
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    g, X, refs = _graph_and_refs()
    n = g.num_nodes
    agg = ShardedAggregator(g.row_pointers, g.column_index, [0, n], 32, device=dev, pipeline_chunks=chunks,
                            exchange=exchange, force_collectives=True)
    assert agg.force_collectives and agg.collectives and agg.overlap and agg.chunks == chunks and agg.exchange == exchange
    assert agg.local_part[0].numel() > 0 and agg.remote_part[0].numel() > 0
    Xd, degd = X.to(dev), g.degrees.to(dev)
    worst = 0.0
    for rep in range(3):                                        # buffers are reused from step to step
        for mode in (0, 1, 2):
            ref, scale, eps = refs[mode]
            _poison_receive_buffers(agg)                        # whatever the remote part reads must have ARRIVED
            y = agg.aggregate(Xd, mode, degrees_local=degd, epsilon=eps)
            err = np.abs(y.cpu().numpy() - ref) / scale
            assert not np.isnan(err).any(), (rep, mode, "a row was read before its collective had delivered it")
            worst = max(worst, float(err.max()))
    assert worst <= 1e-4, worst
    # the two halves of a step on their own (what bench.py reports as exchange_only_ms / aggregate_only_ms)
    _poison_receive_buffers(agg)
    agg.exchange_only(Xd)
    y = agg.aggregate_only(Xd)
    assert float((np.abs(y.cpu().numpy() - refs[0][0]) / refs[0][1]).max()) <= 1e-4
    assert agg.bytes_received_per_step(64) > 0


@pytest.mark.parametrize("chunks,exchange", [(3, "allgather"), (2, "halo")])
def test_the_step_is_ordered_on_a_side_stream(rccl_group, chunks, exchange):
    """The same step issued inside a non-default stream: the collective must wait for that stream's producers (the features
    are written on it just before), and the library's launches on it must wait for the collective."""
    from gnnadvisor_osdi21_amd.dist import ShardedAggregator
    dev = rccl_group
    g, X, refs = _graph_and_refs(seed=43)
    agg = ShardedAggregator(g.row_pointers, g.column_index, [0, g.num_nodes], 32, device=dev, pipeline_chunks=chunks,
                            exchange=exchange, force_collectives=True)
    Xd = X.to(dev)
    side = torch.cuda.Stream(device=dev)
    ref, scale, _ = refs[0]
    torch.cuda.synchronize()
    for rep in range(3):
        with torch.cuda.stream(side):
            _poison_receive_buffers(agg)
            Xs = torch.empty_like(Xd)
            big = torch.randn(4096, 4096, device=dev)
            for _ in range(3):
                big = big @ big * 1e-4                            # keeps the stream busy ahead of the producer
            Xs.copy_(Xd * (1.0 + 0.0 * big[0, 0]))               # the features are produced ON the side stream
            y = agg.sag(Xs)
        side.synchronize()
        err = np.abs(y.cpu().numpy() - ref) / scale
        assert not np.isnan(err).any() and float(err.max()) <= 1e-4, (rep, float(np.nanmax(err)))


def test_sharded_layers_all_reduce_the_weight_gradient_over_rccl(rccl_group):
    """One training step of ShardedGCNConv / ShardedGINConv with the one-rank group: weights broadcast, dW all-reduced
    through RCCL; equal to the single-GPU op layer on the same graph."""
    from gnnadvisor_osdi21_amd.dist import ShardedAggregator, ShardedGCNConv, ShardedGINConv
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import gcn_gin_reference
    dev = rccl_group
    g, _, _ = _graph_and_refs(n=6000, e=300000, D=8, seed=47)
    agg = ShardedAggregator(g.row_pointers, g.column_index, [0, g.num_nodes], 32, device=dev, exchange="auto",
                            pipeline_chunks=2, force_collectives=True)
    l1, l2 = ShardedGCNConv(12, 8, agg), ShardedGINConv(8, 5, agg)
    F = torch.randn(g.num_nodes, 12, generator=torch.Generator().manual_seed(3))
    Fd = F.to(dev).requires_grad_(True)
    degd = g.degrees.to(dev)
    wgt = torch.linspace(0.5, 1.5, 5, device=dev)
    h1 = l1(Fd, degd)
    y = l2(torch.relu(h1))
    (y * wgt).sum().backward()
    ref = gcn_gin_reference(g, F, l1.weights, l2.weights, wgt, H1_got=h1)     # (relu' of a cancelling element: the computed sign)
    assert ref["ambiguous"] <= 16 + 1e-3 * h1.numel()
    for got, (want, scale) in ((y, ref["out"]), (h1, ref["H1"]), (Fd.grad, ref["dF"]), (l1.weights.grad, ref["dW1"]), (l2.weights.grad, ref["dW2"])):
        err = np.abs(got.detach().double().cpu().numpy() - want) / np.maximum(1.0, scale)
        assert float(err.max()) <= 1e-4


def test_bench_runs_the_sharded_path_over_rccl_with_one_rank():
    """`bench.py --gpus 1 --backend nccl --force-collectives`: the N-rank bench path (init_process_group("nccl",
    device_id=...), exchange + kernels, exchange-only / kernels-only timings, verification) on one GPU."""
    import json
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--backend", "nccl", "--force-collectives",
           "--scale", "0.05", "--steps", "3", "--warmup", "1", "--exchange", "halo", "--pipeline-chunks", "2",
           "--headline-only", "--scaling", ""]
    env = dict(os.environ, MASTER_PORT=str(_free_port()))
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    c = rec["config"]
    assert c["rccl_world_size"] == 1 and c["backend"] == "nccl" and c["force_collectives"] is True
    assert c["exchange"] == "halo" and c["exchange_only_ms"] > 0 and c["aggregate_only_ms"] > 0
    assert rec["verified"] is True and rec["value"] > 0
