"""Two ranks sharing the one GPU of the test box (gloo moves the device tensors): the whole
multi-rank path of gnnadvisor_osdi21_amd/dist.py -- sharding, padded / sub-block-major layouts,
asynchronous (optionally K-piece) all-gather, local + windowed remote aggregation -- with the
real HIP kernels as the local aggregation, checked against the oracle on the full graph.
(RCCL itself needs one GPU per rank; `bench.py --gpus N` covers it on a multi-GPU node.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, chunks, q, exchange="allgather", backend="gloo"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    device_index = rank if backend == "nccl" else 0          # RCCL: one GPU per rank; gloo: both ranks share the box's GPU
    if backend == "nccl":
        torch.cuda.set_device(device_index)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import oracle
        from gnnadvisor_osdi21_amd import _lib, graph
        from gnnadvisor_osdi21_amd.dist import (ShardedAggregator, ShardedGCNConv, ShardedGINConv,
                                                balanced_row_splits, shard_csr)
        torch.cuda.set_device(device_index)
        n, e, D, ps = 5000, 400000, 64, 32
        if exchange == "halo":      # low degree, id-local: few of the peer's rows are referenced at all
            n, e = 20000, 200000
        g = graph.powerlaw_graph(n, e, 1500, seed=5, locality=0.97 if exchange == "halo" else 0.0, window=300)
        bounds = balanced_row_splits(g.row_pointers, world)
        lo, hi = bounds[rank], bounds[rank + 1]
        rp, ci = shard_csr(g.row_pointers, g.column_index, lo, hi)
        agg = ShardedAggregator(rp, ci, bounds, ps, device="cuda", pipeline_chunks=chunks, exchange=exchange)
        assert agg.overlap and agg.chunks == chunks and agg.exchange == exchange
        if exchange == "halo":      # id-local graph: the halo is a small part of the peer's block
            assert agg.bytes_received_per_step(D) < 0.5 * agg.allgather_bytes_per_step(D)
        X = torch.randn(n, D, generator=torch.Generator().manual_seed(8))
        Xl = X[lo:hi].contiguous().cuda()
        degl = g.degrees[lo:hi].contiguous().cuda()
        rpn, cin, degn, Xn = g.row_pointers.numpy(), g.column_index.numpy(), g.degrees.numpy(), X.numpy()
        ok, worst, bad = True, 0.0, []
        for rep in range(3):                                    # buffers are reused from step to step
            for mode, eps in ((0, 1.0), (1, 1.0), (2, 0.5)):
                y = agg.aggregate(Xl, mode, degrees_local=degl, epsilon=eps)
                ref = oracle.csr_f64(mode, Xn, rpn, cin, degn, eps)[lo:hi]
                # SAG / GIN: strict |err| <= 1e-4 * max(1, |ref|); the degree-weighted form: 1e-4 of the sum of |terms|
                scale = np.maximum(1.0, oracle.csr_f64(mode, np.abs(Xn), rpn, cin, degn, eps)[lo:hi] if mode == 1 else np.abs(ref))
                e_rows = (np.abs(y.cpu().numpy() - ref) / scale).max(axis=1)
                err = float(e_rows.max())
                worst = max(worst, err)
                ok &= err <= 1e-4
                if err > 1e-4:
                    bad.append(("aggregate rep %d mode %d" % (rep, mode), err, int((e_rows > 1e-4).sum()),
                                np.nonzero(e_rows > 1e-4)[0][:8].tolist()))
        # one training step of the sharded layers against the fp64 network on the whole graph.  STRICT: every intermediate of
        # the step is traced (dist.set_trace: clones + stream / thread ids), a mismatch dumps the lot under
        # gpurun_out/dist_dump/ and fails -- no second attempt (the re-run of rounds 3-4 hid a defect of the CHECKER: the fp64
        # ReLU mask against the fp32 sign of a pre-activation that cancels to ~0, see util.gcn_gin_reference)
        from gnnadvisor_osdi21_amd import dist as gdist
        trace = []
        gdist.set_trace(trace)
        if os.environ.get("GNNA_TEST_MAIN_THREAD_BACKWARD", "0") == "1":
            torch.autograd.set_multithreading_enabled(False)
        l1, l2 = ShardedGCNConv(12, 8, agg), ShardedGINConv(8, 5, agg)
        F = torch.randn(n, 12, generator=torch.Generator().manual_seed(3))
        Fl = F[lo:hi].contiguous().cuda().requires_grad_(True)
        h1 = l1(Fl, degl)
        yl = l2(torch.relu(h1))
        wgt = torch.linspace(0.5, 1.5, 5, device="cuda")
        (yl * wgt).sum().backward()
        gdist.set_trace(None)

        # within 1e-4 of the sum of |terms| (north_star's bound); the replicated weight gradients are complete on every
        # rank (all-reduced), out / H1 / dF are this rank's block.  The reference's ReLU mask follows the computed sign where
        # the pre-activation is inside that bound of zero, so it needs the H1 blocks of ALL ranks.
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
        from util import gcn_gin_reference
        blocks = [None] * world
        dist.all_gather_object(blocks, (lo, h1.detach().cpu()))
        H1_all = torch.cat([b[1] for b in sorted(blocks, key=lambda b: b[0])])
        ref = gcn_gin_reference(g, F, l1.weights, l2.weights, wgt, H1_got=H1_all)
        assert ref["ambiguous"] <= 16 + 1e-3 * H1_all.numel(), ref["ambiguous"]   # (about one element per 30,000 cancels that far)
        for what, got, (want, scale), sl in (("layers out", yl, ref["out"], slice(lo, hi)), ("layers H1", h1, ref["H1"], slice(lo, hi)),
                                             ("layers dF", Fl.grad, ref["dF"], slice(lo, hi)),
                                             ("layers dW1", l1.weights.grad, ref["dW1"], slice(None)),
                                             ("layers dW2", l2.weights.grad, ref["dW2"], slice(None))):
            err = np.abs(got.detach().double().cpu().numpy() - want[sl]) / np.maximum(1.0, scale[sl])
            worst = max(worst, float(err.max()))
            ok &= bool(err.max() <= 1e-4)
            if not err.max() <= 1e-4:
                bad.append((what, float(err.max()), int((err > 1e-4).sum())))
        info = dict(ambiguous=ref["ambiguous"], min_ratio=ref["min_ratio"], sign_flips=ref["sign_flips"])
        if not ok:
            root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            ddir = os.path.join(root, "gpurun_out", "dist_dump")
            os.makedirs(ddir, exist_ok=True)
            path = os.path.join(ddir, "two_ranks_%s_k%d_port%d_rank%d.pt" % (exchange, chunks, port, rank))
            torch.save(dict(exchange=exchange, chunks=chunks, rank=rank, lo=lo, hi=hi, bad=bad, info=info, F=F, wgt=wgt.cpu(),
                            W1=l1.weights.detach().cpu(), W2=l2.weights.detach().cpu(), H1_all=H1_all, out=yl.detach().cpu(),
                            dF=Fl.grad.cpu(), dW1=l1.weights.grad.cpu(), dW2=l2.weights.grad.cpu(),
                            trace=[(name, t.cpu(), stream, thread) for name, t, stream, thread in trace],
                            env={k: v for k, v in os.environ.items() if k.startswith(("GNNA", "AMD_", "HIP", "TORCH_BLAS", "HSA"))}), path)
            info["dump"] = path
        q.put((rank, bool(ok), (worst, bad, info)))
    except Exception as exc:                                   # surface the failure instead of a queue timeout
        import traceback
        q.put((rank, False, traceback.format_exc()))
        raise exc
    finally:
        dist.destroy_process_group()


def _two_ranks_once(chunks, exchange, backend="gloo"):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, chunks, q, exchange, backend)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    codes = []
    for p in procs:
        p.join(timeout=60)
        codes.append(p.exitcode)
    return res, codes


@pytest.mark.parametrize("chunks,exchange", [(1, "allgather"), (3, "allgather"), (1, "halo"), (3, "halo")])
def test_two_ranks_sharing_the_gpu(chunks, exchange):
    """One attempt, strict.  (Rounds 3-4 re-ran a mismatch of the input gradient once: 4 of ~270 sessions had 3-15 rows of one
    rank's dF off while out / dW1 / dW2 were fine.  Cause, named in round 5: the checker's fp64 ReLU mask against the fp32
    sign of a layer-1 pre-activation that cancels to within rounding of zero -- the sign then depends on the order float
    atomics land in; tests/test_relu_mask.py reproduces it deterministically.  The reference now takes the computed sign for
    the elements inside the bound, and a mismatch dumps every intermediate of the step.)"""
    res, codes = _two_ranks_once(chunks, exchange)
    assert codes == [0, 0], (codes, res)
    assert all(ok for _, ok, _ in res), res


@pytest.mark.parametrize("chunks,exchange", [(1, "allgather"), (3, "allgather"), (1, "halo"), (3, "halo")])
def test_two_ranks_over_rccl_on_two_gpus(chunks, exchange):
    """The same strict step with a REAL two-rank RCCL group, one GPU per rank (all_gather_into_tensor / all_to_all_single
    over xGMI, dW all-reduced, weights broadcast).  Skipped on a one-GPU box: it executes the moment the suite runs on a
    multi-GPU node (VERDICT r4 task 6a)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL takes one device per rank)")
    res, codes = _two_ranks_once(chunks, exchange, backend="nccl")
    assert codes == [0, 0], (codes, res)
    assert all(ok for _, ok, _ in res), res


@pytest.mark.parametrize("model", ["gcn", "gin"])
def test_sharded_training_driver_two_ranks(model):
    """python -m torch.distributed.run ... -m gnnadvisor_osdi21_amd.dist_main: epochs run, the loss
    is finite, replicated weights stay bit-identical, the reference's metric line is printed."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           "-m", "gnnadvisor_osdi21_amd.dist_main", "--synthetic", "reddit-like", "--scale", "0.03",
           "--dim", "40", "--hidden", "16", "--classes", "5", "--model", model, "--num_epoches", "3",
           "--backend", "gloo", "--share_gpu", "--pipeline_chunks", "2", "--verbose_mode", "True"]
    res = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600)
    if res.returncode != 0:                                     # one attempt; keep the whole evidence of a failure
        ddir = os.path.join(root, "gpurun_out", "dist_dump")
        os.makedirs(ddir, exist_ok=True)
        with open(os.path.join(ddir, "driver_%s_rc%d_pid%d.log" % (model, res.returncode, os.getpid())), "w") as f:
            f.write(" ".join(cmd) + "\n--- stdout\n" + res.stdout + "\n--- stderr\n" + res.stderr)
    assert res.returncode == 0, res.stderr[-2000:]
    out = res.stdout
    assert re.search(r"Time \(ms\): \d+\.\d{3}", out), out
    assert "# weights identical on all ranks: True" in out, out
    loss = float(re.search(r"# final loss: (-?\d+\.\d+|nan|inf)", out).group(1))
    assert loss == loss and abs(loss) < 1e6


# ---- round 6: exchange="auto" by measurement, with the real kernels -----------------------------------------------------
def _timed_worker(rank, world, port, q, backend="gloo"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    device_index = rank if backend == "nccl" else 0
    if backend == "nccl":
        torch.cuda.set_device(device_index)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import oracle
        from gnnadvisor_osdi21_amd import graph
        from gnnadvisor_osdi21_amd.dist import balanced_row_splits, shard_csr, timed_aggregator
        torch.cuda.set_device(device_index)
        n, e, D, ps = 20000, 600000, 32, 32
        g = graph.powerlaw_graph(n, e, 1500, seed=9, locality=0.9, window=300)
        bounds = balanced_row_splits(g.row_pointers, world)
        lo, hi = bounds[rank], bounds[rank + 1]
        rp, ci = shard_csr(g.row_pointers, g.column_index, lo, hi)
        agg = timed_aggregator(rp, ci, bounds, ps, dim=D, reps=3, device="cuda")
        rec = agg.exchange_timed
        X = torch.randn(n, D, generator=torch.Generator().manual_seed(10))
        y = agg.sag(X[lo:hi].contiguous().cuda())
        ref = oracle.csr_f64(0, X.numpy(), g.row_pointers.numpy(), g.column_index.numpy())[lo:hi]
        err = np.abs(y.cpu().numpy() - ref) / np.maximum(1.0, np.abs(ref))
        q.put((rank, bool(err.max() <= 1e-4), agg.exchange, rec))
    except Exception as exc:
        import traceback
        q.put((rank, False, traceback.format_exc(), None))
        raise exc
    finally:
        dist.destroy_process_group()


def _timed_once(backend):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_timed_worker, args=(r, 2, port, q, backend)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0, res
    return res


def _check_timed(res):
    assert all(ok for _, ok, _, _ in res), res
    (_, _, ex0, rec0), (_, _, ex1, rec1) = sorted(res)
    assert ex0 == ex1 == rec0["chosen"] and rec0 == rec1            # one choice, from the same two numbers, on both ranks
    assert rec0["allgather_ms"] > 0 and rec0["halo_ms"] > 0
    assert rec0[rec0["chosen"] + "_ms"] == min(rec0["allgather_ms"], rec0["halo_ms"])


def test_exchange_chosen_by_timing_with_the_real_kernels():
    """dist.timed_aggregator on two ranks sharing the GPU (gloo): both forms are built on the device, each runs three whole steps
    with the HIP kernels, the faster is kept on both ranks and gives the whole graph's result."""
    _check_timed(_timed_once("gloo"))


def test_exchange_chosen_by_timing_over_rccl_on_two_gpus():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL takes one device per rank)")
    _check_timed(_timed_once("nccl"))
