"""Multi-process sharded aggregation on CPU (gloo, world_size 2): the sharding, padding,
column remap and all-gather logic of gnnadvisor_osdi21_amd/dist.py.  The local kernel is
the injectable ``aggregate_fn``; here (and only here) the oracle stands in for it as the
checker, since there is no GPU in this container."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from gnnadvisor_osdi21_amd import graph
from gnnadvisor_osdi21_amd.dist import (ShardedAggregator, balanced_row_splits, remap_columns_to_padded,
                                        shard_csr, split_local_remote)


def _oracle_aggregate(mode, X_all, column_index, part_pointers, part2Node, num_out_rows, partSize,
                      degrees_out=None, degrees_in=None, epsilon=1.0, out=None, accumulate=False, windows=None):
    X = X_all.numpy(); ci = column_index.numpy(); pp = part_pointers.numpy(); p2n = part2Node.numpy()
    dim = X.shape[1]
    Y = np.zeros((num_out_rows, dim), dtype=np.float32)
    lo_id, hi_id = 0, X.shape[0]
    if windows is not None:                     # (K, begin, end): only sources inside these windows
        K, wb, we = windows
        wrows = (X.shape[0] + K - 1) // K
        lo_id, hi_id = wb * wrows, we * wrows
        accumulate = accumulate or wb > 0
    for p in range(len(p2n)):
        r = p2n[p]
        for e in range(pp[p], pp[p + 1]):
            if not (lo_id <= ci[e] < hi_id):
                continue
            c = 1.0
            if mode == 1:
                c = np.float32(degrees_out[r].item()) * np.float32(degrees_in[ci[e]].item())
            Y[r] += np.float32(c) * X[ci[e]]
    if mode == 2:
        Y *= np.float32(epsilon)
    res = torch.from_numpy(Y)
    if out is not None:
        if accumulate:
            out.add_(res)
        else:
            out.copy_(res)
        return out
    assert not accumulate
    return res


def _oracle_build_part(ps, rp):
    pp, p2n = oracle.build_part(ps, rp.numpy())
    return torch.from_numpy(pp), torch.from_numpy(p2n)


def _worker(rank, world, port, n, e, dim, seed, q, overlap=True, chunks=1, exchange="allgather", locality=0.0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = graph.powerlaw_graph(n, e, 60, seed=seed, locality=locality, window=6)   # same graph on every rank
        bounds = balanced_row_splits(g.row_pointers, world)
        lo, hi = bounds[rank], bounds[rank + 1]
        rp, ci = shard_csr(g.row_pointers, g.column_index, lo, hi)
        X = torch.randn(n, dim, generator=torch.Generator().manual_seed(seed + 1))
        agg = ShardedAggregator(rp, ci, bounds, 4, aggregate_fn=_oracle_aggregate,
                                build_part_fn=_oracle_build_part, overlap=overlap, pipeline_chunks=chunks,
                                exchange=exchange)
        assert agg.overlap == overlap and agg.chunks == (chunks if overlap else 1)
        if exchange == "halo":
            assert agg.exchange == "halo" and agg.halo_rows <= (world - 1) * agg.rows_per_rank
            assert agg.bytes_received_per_step(dim) == agg.halo_rows * dim * 4
        if overlap:
            assert agg.local_part[0].numel() + agg.remote_part[0].numel() == ci.numel()
            assert bool((agg.local_part[0] >= 0).all()) and bool((agg.local_part[0] < hi - lo).all())
        Ys = agg.sag(X[lo:hi].contiguous())
        Yg = agg.aggregate(X[lo:hi].contiguous(), 1, degrees_local=g.degrees[lo:hi].contiguous())
        Yi = agg.aggregate(X[lo:hi].contiguous(), 2, epsilon=0.5)
        rpn, cin = g.row_pointers.numpy(), g.column_index.numpy()
        ok = True
        ok &= np.allclose(Ys.numpy(), oracle.csr_f64(0, X.numpy(), rpn, cin)[lo:hi], atol=1e-4)
        ok &= np.allclose(Yg.numpy(), oracle.csr_f64(1, X.numpy(), rpn, cin, g.degrees.numpy())[lo:hi], rtol=1e-4, atol=1e-2)
        ok &= np.allclose(Yi.numpy(), oracle.csr_f64(2, X.numpy(), rpn, cin, None, 0.5)[lo:hi], atol=1e-4)
        q.put((rank, bool(ok), lo, hi, agg.rows_per_rank, agg.exchange,
               agg.bytes_received_per_step(dim) / max(1, agg.allgather_bytes_per_step(dim))))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("n,e,overlap,chunks", [(101, 1500, True, 1), (64, 40, True, 1), (101, 1500, False, 1),
                                                 (101, 1500, True, 3), (64, 40, True, 4)])
def test_two_rank_sharded_aggregation_matches_single_graph(n, e, overlap, chunks):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, e, 12, 7, q, overlap, chunks)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, *_ in res), res
    spans = sorted((lo, hi) for _, _, lo, hi, *_ in res)
    assert spans[0][0] == 0 and spans[0][1] == spans[1][0] and spans[1][1] == n   # rows tile exactly


def _run_ranks(target, args, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, *args, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def _halo_worker(rank, world, port, n, e, chunks, exchange, locality, q):
    _worker(rank, world, port, n, e, 10, 13, q, True, chunks, exchange, locality)


@pytest.mark.parametrize("n,e,chunks,locality", [(120, 1800, 1, 0.0), (120, 1800, 3, 0.0), (400, 3000, 1, 0.9),
                                                  (400, 3000, 4, 0.9), (64, 40, 2, 0.0)])
def test_two_rank_halo_exchange_matches_single_graph(n, e, chunks, locality):
    """exchange="halo": only the referenced remote rows travel (all_to_all_single), the remote part reads the
    compact halo buffer; results equal the whole-graph oracle in all three modes, with and without the
    K-piece pipeline.  On an id-local graph the halo is well under half of the all-gather volume."""
    res = _run_ranks(_halo_worker, (n, e, chunks, "halo", locality))
    assert all(ok for _, ok, *_ in res), res
    assert all(r[5] == "halo" for r in res)
    if locality >= 0.9:
        assert all(r[6] < 0.5 for r in res), [r[6] for r in res]


def _auto_worker(rank, world, port, n, e, locality, q):
    _worker(rank, world, port, n, e, 10, 13, q, True, 1, "auto", locality)


def test_auto_exchange_is_a_collective_choice():
    """exchange="auto": the all-gather stays on a graph whose shards reference nearly every remote row, the
    halo exchange is taken on an id-local one -- and both ranks always take the same path."""
    dense = _run_ranks(_auto_worker, (60, 3000, 0.0))
    assert all(ok for _, ok, *_ in dense) and {r[5] for r in dense} == {"allgather"}
    local = _run_ranks(_auto_worker, (400, 3000, 0.9))
    assert all(ok for _, ok, *_ in local) and {r[5] for r in local} == {"halo"}


def _uneven_k_worker(rank, world, port, q):
    """the automatic piece count must be the same on every rank even when their shards differ."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = graph.powerlaw_graph(90, 1200, 40, seed=3)
        bounds = [0, 20, 90]                                   # very different shard sizes
        lo, hi = bounds[rank], bounds[rank + 1]
        rp, ci = shard_csr(g.row_pointers, g.column_index, lo, hi)
        agg = ShardedAggregator(rp, ci, bounds, 4, aggregate_fn=_oracle_aggregate, build_part_fn=_oracle_build_part,
                                pipeline_chunks=0)
        ks = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(ks, torch.tensor([agg.chunks]))
        q.put((rank, len({int(k) for k in ks}) == 1))
    finally:
        dist.destroy_process_group()


def test_automatic_piece_count_is_agreed_between_ranks():
    assert all(ok for _, ok in _run_ranks(_uneven_k_worker, ()))


def test_piece_count_model_on_the_bench_shapes():
    """The automatic piece count of the pipelined exchange (time model, dist.exposed_exchange_us): Reddit-sized
    weak-scaling shards hide the exchange behind the local-source part at 2 ranks and do not at 4 and 8;
    a papers100M-like shard (14.5 edges per row) is exchange-bound at any size."""
    from gnnadvisor_osdi21_amd.dist import exposed_exchange_us
    n, nnz = 232965, 114623790
    assert exposed_exchange_us(2, nnz, n)[0] <= 0
    assert exposed_exchange_us(4, nnz, n)[0] > 0 and exposed_exchange_us(8, nnz, n)[0] > 0
    assert exposed_exchange_us(8, 196094012, 13882494)[0] > 0
    assert exposed_exchange_us(2, 196094012, 13882494)[0] > 0
    assert exposed_exchange_us(1, nnz, n) == (int(-1e6 * nnz / 60e9), 0)


def test_balanced_splits_and_remap():
    rp = torch.tensor([0, 10, 10, 11, 30, 31, 40], dtype=torch.int32)
    b = balanced_row_splits(rp, 2)
    assert b[0] == 0 and b[-1] == 6 and b == sorted(b)
    assert abs(int(rp[b[1]]) - 20) <= 10
    assert balanced_row_splits(rp, 1) == [0, 6]
    b8 = balanced_row_splits(rp, 8)
    assert len(b8) == 9 and b8 == sorted(b8) and b8[-1] == 6
    # padded layout: owner * rows_per_rank + local offset
    ci = torch.tensor([0, 2, 3, 5], dtype=torch.int32)
    out = remap_columns_to_padded(ci, [0, 3, 6], 4)
    assert out.tolist() == [0, 2, 4, 6]
    lrp, lci = shard_csr(rp, torch.arange(40, dtype=torch.int32), 3, 6)
    assert lrp.tolist() == [0, 19, 20, 29] and lci[0] == 11 and lci.numel() == 29


def test_split_local_remote_partitions_every_edge():
    g = graph.powerlaw_graph(200, 3000, 80, seed=4)
    rp, ci = shard_csr(g.row_pointers, g.column_index, 50, 120)
    rp_l, ci_l, rp_r, ci_r = split_local_remote(rp, ci, 50, 120)
    assert ci_l.numel() + ci_r.numel() == ci.numel()
    assert bool(((ci_l >= 0) & (ci_l < 70)).all()) and not bool(((ci_r >= 50) & (ci_r < 120)).any())
    deg = (rp[1:] - rp[:-1])
    assert torch.equal((rp_l[1:] - rp_l[:-1]) + (rp_r[1:] - rp_r[:-1]), deg)
    for r in range(70):   # per row: local + remote ids == original ids
        orig = sorted(ci[rp[r]:rp[r + 1]].tolist())
        got = sorted([v + 50 for v in ci_l[rp_l[r]:rp_l[r + 1]].tolist()] + ci_r[rp_r[r]:rp_r[r + 1]].tolist())
        assert orig == got


def _train_worker(rank, world, port, n, e, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gnnadvisor_osdi21_amd.dist import ShardedGCNConv, ShardedGINConv
        fin, hid, ncls = 9, 6, 4
        g = graph.powerlaw_graph(n, e, 40, seed=11)
        bounds = balanced_row_splits(g.row_pointers, world)
        lo, hi = bounds[rank], bounds[rank + 1]
        rp, ci = shard_csr(g.row_pointers, g.column_index, lo, hi)
        agg = ShardedAggregator(rp, ci, bounds, 3, aggregate_fn=_oracle_aggregate,
                                build_part_fn=_oracle_build_part, overlap=True, pipeline_chunks=2)
        torch.manual_seed(100 + rank)                      # different draws: the broadcast must align them
        l1 = ShardedGCNConv(fin, hid, agg, device="cpu")
        l2 = ShardedGINConv(hid, ncls, agg, device="cpu")
        X = torch.randn(n, fin, generator=torch.Generator().manual_seed(5))
        Xl = X[lo:hi].clone().requires_grad_(True)
        deg = g.degrees[lo:hi].contiguous()
        y = l2(torch.relu(l1(Xl, deg)))
        # loss = sum over ALL nodes: each rank back-propagates its rows' share
        loss_local = (y * torch.arange(1, ncls + 1, dtype=torch.float32)).sum()
        loss_local.backward()

        # single-process dense autograd of the same 2-layer net on the whole graph
        A = torch.zeros(n, n, dtype=torch.float64)
        rows = torch.repeat_interleave(torch.arange(n), (g.row_pointers[1:] - g.row_pointers[:-1]).long())
        A[rows, g.column_index.long()] = 1.0
        d = g.degrees.double()
        Ahat = A * d[:, None] * d[None, :]
        W1 = l1.weights.detach().double().requires_grad_(True)
        W2 = l2.weights.detach().double().requires_grad_(True)
        Xd = X.double().requires_grad_(True)
        yd = (0.5 * A @ torch.relu(Ahat @ (Xd @ W1))) @ W2
        (yd * torch.arange(1, ncls + 1, dtype=torch.float64)).sum().backward()
        ok = True
        ok &= torch.allclose(y.double(), yd[lo:hi].detach(), rtol=1e-4, atol=1e-3)
        ok &= torch.allclose(l1.weights.grad.double(), W1.grad, rtol=1e-4, atol=1e-2)
        ok &= torch.allclose(l2.weights.grad.double(), W2.grad, rtol=1e-4, atol=1e-2)
        ok &= torch.allclose(Xl.grad.double(), Xd.grad[lo:hi], rtol=1e-4, atol=1e-2)
        # replicated weights stayed replicated
        w_all = [torch.empty_like(l1.weights.data) for _ in range(world)]
        dist.all_gather(w_all, l1.weights.data)
        ok &= all(torch.equal(w_all[0], w) for w in w_all)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_layers_match_dense_autograd():
    """GCN then GIN layer on a 2-way sharded graph (gloo): outputs, dX and the all-reduced dW equal
    single-process dense autograd on the whole graph (A symmetric, as the reference assumes)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, 83, 900, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


@pytest.mark.parametrize("exchange,chunks,overlap", [("allgather", 1, True), ("allgather", 3, True), ("halo", 1, True),
                                                      ("halo", 2, True), ("allgather", 1, False)])
def test_one_process_emulation_of_every_rank_matches_the_whole_graph(exchange, chunks, overlap):
    """`emulate=(rank, world)`: one process builds rank r's shard, split, halo lists and piece CSRs exactly as a
    `world`-rank job would and gets its receive buffer from `emulated_receive` instead of a collective (how bench.py
    and the GPU tests run BASELINE config 5's per-rank shape on one GPU).  Every rank's rows must equal the rows of
    the single-graph result, in all three modes."""
    n, e, dim, world = 157, 2600, 10, 3
    g = graph.powerlaw_graph(n, e, 60, seed=9, locality=0.5, window=6)
    bounds = balanced_row_splits(g.row_pointers, world)
    X = torch.randn(n, dim, generator=torch.Generator().manual_seed(10))
    rpn, cin = g.row_pointers.numpy(), g.column_index.numpy()
    want = [oracle.csr_f64(0, X.numpy(), rpn, cin), oracle.csr_f64(1, X.numpy(), rpn, cin, g.degrees.numpy()),
            oracle.csr_f64(2, X.numpy(), rpn, cin, None, 0.5)]
    for rank in range(world):
        lo, hi = bounds[rank], bounds[rank + 1]
        rp, ci = shard_csr(g.row_pointers, g.column_index, lo, hi)
        agg = ShardedAggregator(rp, ci, bounds, 4, aggregate_fn=_oracle_aggregate, build_part_fn=_oracle_build_part,
                                overlap=overlap, force_overlap=overlap, pipeline_chunks=chunks, exchange=exchange,
                                emulate=(rank, world))
        assert agg.emulated and agg.world == world and agg.rank == rank and agg.exchange == exchange
        if exchange == "halo":
            assert sum(agg.halo_rows_per_peer) == agg.halo_rows and agg.halo_rows_per_peer[rank] == 0
            sent = agg.send_side_gather(X[lo:hi].contiguous())          # stand-in send lists: as many rows as are received
            assert len(sent) == agg.chunks and sum(t.shape[0] for t in sent) == agg.halo_rows
        buf = agg.emulated_receive(X)
        assert buf.shape[0] == (agg.remote_rows if overlap else world * agg.rows_per_rank)
        agg.emulated_receive_degrees(g.degrees)
        Xl, dl = X[lo:hi].contiguous(), g.degrees[lo:hi].contiguous()
        Ys = agg.aggregate_only(Xl)
        Yg = agg.aggregate_only(Xl, mode=1, degrees_local=dl)
        Yi = agg.aggregate_only(Xl, mode=2, epsilon=0.5)
        assert np.allclose(Ys.numpy(), want[0][lo:hi], atol=1e-4), (rank, "sag")
        assert np.allclose(Yg.numpy(), want[1][lo:hi], rtol=1e-4, atol=1e-2), (rank, "gcn")
        assert np.allclose(Yi.numpy(), want[2][lo:hi], atol=1e-4), (rank, "gin")


def _worker_file(rank, world, port, path, n, dim, seed, q, reorder):
    """Two ranks ingest the same .npz: each builds only its own destination rows (loader.load_graph_shard) and
    aggregates through the sharded path; with `reorder` the node renumbering is computed by rank 0 and broadcast."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gnnadvisor_osdi21_amd.loader import load_graph_shard

        def share(t, src):
            dist.broadcast(t, src=src)
            return t
        sh = load_graph_shard(path, rank, world, reorder=reorder, share_fn=share)
        lo, hi = sh.row_range
        X_old = torch.randn(n, dim, generator=torch.Generator().manual_seed(seed + 1))   # features by ORIGINAL node id
        X = X_old
        if reorder:
            X = torch.empty_like(X_old)
            X[sh.new_id.long()] = X_old                                                 # row new_id[i] holds node i
        agg = ShardedAggregator(sh.row_pointers, sh.column_index, sh.bounds, 4, aggregate_fn=_oracle_aggregate,
                                build_part_fn=_oracle_build_part, exchange="auto")
        Y = agg.sag(X[lo:hi].contiguous())
        Yg = agg.aggregate(X[lo:hi].contiguous(), 1, degrees_local=sh.degrees)
        q.put((rank, lo, hi, Y.numpy(), Yg.numpy(), None if sh.new_id is None else sh.new_id.numpy(), agg.exchange))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("reorder", [False, True])
def test_two_ranks_ingest_a_graph_file_and_match_the_single_process_result(tmp_path, reorder):
    n, dim, seed = 180, 6, 21
    g = graph.powerlaw_graph(n, 3000, 50, seed=seed, locality=0.8, window=8)
    rows = torch.repeat_interleave(torch.arange(n), (g.row_pointers[1:] - g.row_pointers[:-1]).long()).numpy()
    perm = np.random.default_rng(3).permutation(n)                   # scramble the ids: something to renumber
    src, dst = perm[rows], perm[g.column_index.numpy()]
    path = str(tmp_path / "graph.npz")
    np.savez(path, src_li=src, dst_li=dst, num_nodes=n)
    # single-process reference on the file's own numbering (loader semantics: dedup + sort)
    from gnnadvisor_osdi21_amd import _lib
    rp1, ci1 = _lib.csr_from_edges(src, dst, n)
    deg1 = _lib.degrees(rp1)
    X_old = torch.randn(n, dim, generator=torch.Generator().manual_seed(seed + 1))
    want = oracle.csr_f64(0, X_old.numpy(), rp1.numpy(), ci1.numpy())
    want_g = oracle.csr_f64(1, X_old.numpy(), rp1.numpy(), ci1.numpy(), deg1.numpy())
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_file, args=(r, 2, port, path, n, dim, seed, q, reorder)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    covered = 0
    for rank, lo, hi, Y, Yg, new_id, exchange in res:
        # row r of the shard is node `old` with new_id[old] == lo + r (identity without renumbering)
        old = np.arange(lo, hi) if new_id is None else np.argsort(new_id)[lo:hi]
        assert np.allclose(Y, want[old], atol=1e-4), rank
        assert np.allclose(Yg, want_g[old], rtol=1e-4, atol=1e-2), rank
        covered += hi - lo
    assert covered == n


def _forced_worker(rank, world, port, chunks, exchange, q):
    """ONE rank, every collective of the N-rank step still issued through the process group (the bring-up switch for
    RCCL on a single GPU, here over gloo): the second half of the rank's own block travels through all_gather_into_tensor
    / all_to_all_single to the rank itself before the remote part reads it."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, e, dim = 150, 2400, 9
        g = graph.powerlaw_graph(n, e, 60, seed=31)
        bounds = [0, n]
        X = torch.randn(n, dim, generator=torch.Generator().manual_seed(32))
        calls = {"all_gather": 0, "all_to_all": 0, "all_reduce": 0}
        real = (dist.all_gather_into_tensor, dist.all_to_all_single, dist.all_reduce)

        def counted(name, fn):
            def f(*a, **k):
                calls[name] += 1
                return fn(*a, **k)
            return f
        dist.all_gather_into_tensor = counted("all_gather", real[0])
        dist.all_to_all_single = counted("all_to_all", real[1])
        dist.all_reduce = counted("all_reduce", real[2])
        try:
            agg = ShardedAggregator(g.row_pointers, g.column_index, bounds, 4, aggregate_fn=_oracle_aggregate,
                                    build_part_fn=_oracle_build_part, pipeline_chunks=chunks, exchange=exchange,
                                    force_collectives=True)
            assert agg.force_collectives and agg.collectives and agg.overlap and agg.chunks == chunks
            assert agg.exchange == exchange
            # both halves carry edges, and together they are the shard
            assert agg.local_part[0].numel() > 0 and agg.remote_part[0].numel() > 0
            assert agg.local_part[0].numel() + agg.remote_part[0].numel() == g.column_index.numel()
            assert int(agg.local_part[0].max()) < n // 2
            rpn, cin = g.row_pointers.numpy(), g.column_index.numpy()
            ok = True
            for rep in range(3):                       # buffers are reused from step to step
                Ys = agg.sag(X)
                Yg = agg.aggregate(X, 1, degrees_local=g.degrees)
                Yi = agg.aggregate(X, 2, epsilon=0.5)
                ok &= np.allclose(Ys.numpy(), oracle.csr_f64(0, X.numpy(), rpn, cin), atol=1e-4)
                ok &= np.allclose(Yg.numpy(), oracle.csr_f64(1, X.numpy(), rpn, cin, g.degrees.numpy()), rtol=1e-4, atol=1e-2)
                ok &= np.allclose(Yi.numpy(), oracle.csr_f64(2, X.numpy(), rpn, cin, None, 0.5), atol=1e-4)
            agg.exchange_only(X)
            y2 = agg.aggregate_only(X)
            ok &= np.allclose(y2.numpy(), oracle.csr_f64(0, X.numpy(), rpn, cin), atol=1e-4)
            # the collectives really ran: per aggregation `chunks` exchanges of the step's kind
            kind = "all_to_all" if exchange == "halo" else "all_gather"
            ok &= calls[kind] >= 10 * chunks
            ok &= calls["all_reduce"] >= 1                      # the set-up decisions
            # sharded layers: dW all-reduced through the group even with one rank
            from gnnadvisor_osdi21_amd.dist import ShardedGCNConv
            before = calls["all_reduce"]
            layer = ShardedGCNConv(dim, 4, agg, device="cpu")
            Xg = X.clone().requires_grad_(True)
            layer(Xg, g.degrees).sum().backward()
            ok &= calls["all_reduce"] == before + 1 and bool(torch.isfinite(layer.weights.grad).all())
        finally:
            dist.all_gather_into_tensor, dist.all_to_all_single, dist.all_reduce = real
        q.put((rank, bool(ok), dict(calls)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("chunks,exchange", [(1, "allgather"), (3, "allgather"), (1, "halo"), (2, "halo")])
def test_one_rank_with_forced_collectives_matches_the_whole_graph(chunks, exchange):
    res = _run_ranks(_forced_worker, (chunks, exchange), world=1)
    assert all(ok for _, ok, *_ in res), res


# ---- round 6: exchange="auto" decided by measurement (dist.timed_aggregator) -----------------------------------------
def _timed_worker(rank, world, port, slow_form, q):
    """Both forms are built and timed; an injected aggregate_fn makes `slow_form` the slower one (it sleeps), whatever the
    bytes say -- the faster form must be kept on BOTH ranks, and it must still give the whole-graph result."""
    import time
    from gnnadvisor_osdi21_amd.dist import timed_aggregator
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, e, dim = 400, 3000, 6
        g = graph.powerlaw_graph(n, e, 60, seed=13, locality=0.9, window=6)       # id-local: the byte rule says "halo"
        bounds = balanced_row_splits(g.row_pointers, world)
        lo, hi = bounds[rank], bounds[rank + 1]
        rp, ci = shard_csr(g.row_pointers, g.column_index, lo, hi)
        state = {"form": None}

        def build_part(ps, rp_):
            return _oracle_build_part(ps, rp_)

        def aggregate(mode, X_all, column_index, *a, **k):
            # the halo form's remote part reads a buffer of halo_rows rows, the all-gather form's one of world * rows_per_rank
            if state["form"] == slow_form and rank == 1:                          # ONE slow rank is enough: MAX over ranks counts
                time.sleep(0.02)
            return _oracle_aggregate(mode, X_all, column_index, *a, **k)
        # (the factory builds the all-gather form first, then the halo form)
        order = iter(("allgather", "halo"))
        real_init = ShardedAggregator.__init__

        def tracking_init(self, *a, **k):
            state["form"] = k.get("exchange")
            real_init(self, *a, **k)
        ShardedAggregator.__init__ = tracking_init
        try:
            calls = {"allgather": 0, "halo": 0}

            def counting(mode, X_all, column_index, *a, **k):
                calls[state["form"]] += 1
                return aggregate(mode, X_all, column_index, *a, **k)
            agg = timed_aggregator(rp, ci, bounds, 4, dim=dim, reps=3, aggregate_fn=counting, build_part_fn=build_part)
        finally:
            ShardedAggregator.__init__ = real_init
        rec = agg.exchange_timed
        state["form"] = agg.exchange
        X = torch.randn(n, dim, generator=torch.Generator().manual_seed(14))
        Y = agg.sag(X[lo:hi].contiguous())
        ok = np.allclose(Y.numpy(), oracle.csr_f64(0, X.numpy(), g.row_pointers.numpy(), g.column_index.numpy())[lo:hi], atol=1e-4)
        # the byte rule alone would have said halo on this graph
        byte_rule = ShardedAggregator(rp, ci, bounds, 4, aggregate_fn=_oracle_aggregate, build_part_fn=_oracle_build_part,
                                      exchange="auto").exchange
        q.put((rank, bool(ok), agg.exchange, rec, byte_rule, calls["allgather"] > 0 and calls["halo"] > 0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("slow_form", ["halo", "allgather"])
def test_exchange_is_chosen_by_measurement_when_there_is_a_wire(slow_form):
    res = _run_ranks(_timed_worker, (slow_form,))
    fast = "allgather" if slow_form == "halo" else "halo"
    for rank, ok, chosen, rec, byte_rule, both_ran in res:
        assert ok and both_ran and chosen == fast == rec["chosen"], (rank, chosen, rec)
        assert rec[slow_form + "_ms"] > rec[fast + "_ms"] and rec["reps"] == 3 and rec["dim"] == 6
        assert byte_rule == "halo"                       # (what "auto" by bytes picks here, whichever form measured faster)
    assert res[0][3] == res[1][3]                        # the same numbers on both ranks: the slowest rank's time


def test_timed_choice_falls_back_to_the_byte_rule_without_a_wire():
    from gnnadvisor_osdi21_amd.dist import timed_aggregator
    g = graph.powerlaw_graph(200, 2000, 40, seed=5)
    bounds = balanced_row_splits(g.row_pointers, 4)
    rp, ci = shard_csr(g.row_pointers, g.column_index, bounds[1], bounds[2])
    agg = timed_aggregator(rp, ci, bounds, 4, dim=8, aggregate_fn=_oracle_aggregate, build_part_fn=_oracle_build_part, emulate=(1, 4))
    assert agg.exchange_timed is None and agg.exchange in ("allgather", "halo")
