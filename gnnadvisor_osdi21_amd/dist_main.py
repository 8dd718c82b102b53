"""Multi-GPU training driver: the reference's epoch loop (GNNA_main.py:143-202 -- 2-layer GCN or
5-layer GIN, Adam lr 0.01, nll_loss of log_softmax, 10 dry runs, ``Time (ms): %.3f``) over a graph
sharded by destination-node range, one process per GPU.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        -m gnnadvisor_osdi21_amd.dist_main --synthetic papers100M-like --model gcn --hidden 128

Every rank generates its own destination block of the named synthetic graph (``num_nodes / world``
rows, ``num_edges / world`` edges, sources drawn from all ranks' nodes -- BASELINE.json config 5
shape at ``--scale 1``), holds the rows of X and the labels of its block, and runs the sharded
layers of ``dist.py``: dense update on the local rows, K-piece RCCL all-gather pipelined with the
per-source-window aggregation, and an all-reduce of the (KB-sized) weight gradients.  The loss is
the mean over ALL nodes (each rank contributes its block's sum / N_global), so the replicated
weights receive identical updates on every rank.  The reference has no multi-GPU mode; this is the
MI355X design of SURVEY.md 8(e).
"""
import argparse
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def build_parser():
    p = argparse.ArgumentParser(description="sharded GCN / GIN training epochs (one process per GPU)")
    p.add_argument("--synthetic", type=str, default="papers100M-like", help="graph.CONFIGS name (used when no --dataset is given)")
    p.add_argument("--dataDir", type=str, default="../osdi-ae-graphs", help="directory of the graph file (GNNA_main.py:20)")
    p.add_argument("--dataset", type=str, default="", help="graph file stem: <dataDir>/<dataset>.npz (keys src_li, dst_li, "
                                                             "num_nodes) or, with --loadFromTxt True, <dataDir>/<dataset> as a "
                                                             "'src dst' text edge list; every rank builds only its own rows")
    p.add_argument("--loadFromTxt", default="False", choices=["True", "False"])
    p.add_argument("--enable_rabbit", default="False", choices=["True", "False"],
                   help="renumber the nodes for locality (native community renumbering, once, on rank 0) before the split")
    p.add_argument("--exchange", default="auto", choices=["auto", "halo", "allgather"])
    p.add_argument("--scale", type=float, default=1.0, help="shrink the whole graph (nodes and edges)")
    p.add_argument("--dim", type=int, default=0, help="input width (0: the config's)")
    p.add_argument("--hidden", type=int, default=0, help="hidden width (0: the config's)")
    p.add_argument("--classes", type=int, default=0, help="classes (0: the config's)")
    p.add_argument("--model", type=str, default="gcn", choices=["gcn", "gin"])
    p.add_argument("--num_epoches", type=int, default=20)
    p.add_argument("--partSize", type=int, default=0, help="0: the Decider's choice")
    p.add_argument("--pipeline_chunks", type=int, default=0, help="pieces of the feature exchange (0: automatic)")
    p.add_argument("--backend", default="nccl", help="debug: gloo")
    p.add_argument("--share_gpu", action="store_true", help="debug: every rank on cuda:0 (with --backend gloo)")
    p.add_argument("--verbose_mode", default="False", choices=["True", "False"])
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    verbose = args.verbose_mode == "True"
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "requires MI355X GPUs: there is no CPU path"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29512")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    import datetime
    kw = dict(timeout=datetime.timedelta(seconds=600))
    if args.backend == "nccl":
        dist.init_process_group("nccl", device_id=dev, **kw)
    else:
        dist.init_process_group(args.backend, **kw)
    if rank == 0:
        print(args)

    from . import graph
    from .decider import choose_part_size
    from .dist import ShardedAggregator, ShardedGCNConv, ShardedGINConv

    cfg = graph.CONFIGS[args.synthetic]
    fin = args.dim or cfg["feat"]
    hidden = args.hidden or cfg["hidden"]
    ncls = args.classes or cfg["classes"]
    if args.dataset:
        # a graph file: every rank reads the edge list and builds only the CSR rows of its nnz-balanced block
        from .loader import load_graph_shard
        txt = args.loadFromTxt == "True"
        path = os.path.join(args.dataDir, args.dataset + ("" if txt else ".npz"))

        def share(t, src_rank):
            t = t.to(dev) if args.backend == "nccl" else t
            dist.broadcast(t, src=src_rank)
            return t.cpu()
        shard = load_graph_shard(path, rank, world, load_from_txt=txt, reorder=args.enable_rabbit == "True",
                                 share_fn=share if world > 1 else None, verbose=verbose)
        rp, ci, bounds = shard.row_pointers.to(dev), shard.column_index.to(dev), shard.bounds
        n_global = shard.num_nodes
        n_local = bounds[rank + 1] - bounds[rank]
    else:
        n_global = max(world * 2, int(cfg["num_nodes"] * args.scale))
        n_local = n_global // world                                   # equal blocks (the generator is uniform over rows)
        n_global = n_local * world
        e_local = int(cfg["num_edges"] * args.scale * cfg.get("oversample", 1.0)) // world
        rp, ci = graph.powerlaw_shard(n_local, n_global, e_local, min(cfg["max_degree"], n_global - 1),
                                      seed=cfg["seed"] * 1000 + rank, device=dev)
        bounds = [i * n_local for i in range(world + 1)]
    avg_degree = ci.numel() / max(1, n_local)
    ps = args.partSize or choose_part_size(avg_degree, hidden)
    if args.exchange == "auto" and world > 1:
        from .dist import timed_aggregator                     # a wire to time: the faster of the two forms, measured
        agg = timed_aggregator(rp, ci, bounds, ps, dim=hidden, reps=3, device=dev, pipeline_chunks=args.pipeline_chunks)
        if verbose and rank == 0:
            print("# exchange (timed): {}".format(agg.exchange_timed))
    else:
        agg = ShardedAggregator(rp, ci, bounds, ps, device=dev, pipeline_chunks=args.pipeline_chunks, exchange=args.exchange)
    # sqrt(max(deg, 1)) of the local rows (dataset.py:121-122; in-degree == out-degree on a symmetric graph)
    deg_local = (rp[1:] - rp[:-1]).clamp(min=1).to(torch.float32).sqrt()
    gen = torch.Generator(device=dev).manual_seed(100 + rank)
    X = torch.randn(n_local, fin, device=dev, generator=gen)       # dataset.py:129
    y = torch.ones(n_local, dtype=torch.long, device=dev)          # dataset.py:136: all-ones labels
    if verbose and rank == 0:
        print(f"# {world} rank(s); per rank: {n_local} rows, {ci.numel()} edges, partSize {ps}, "
              f"exchange in {agg.chunks} piece(s), X {fin} -> {hidden} -> {ncls}")

    torch.manual_seed(0)
    if args.model == "gcn":
        layers = torch.nn.ModuleList([ShardedGCNConv(fin, hidden, agg), ShardedGCNConv(hidden, ncls, agg)])

        def forward():
            h = F.relu(layers[0](X, deg_local))
            return F.log_softmax(layers[1](h, deg_local), dim=1)
    else:
        dims = [fin] + [hidden] * 4 + [ncls]
        layers = torch.nn.ModuleList([ShardedGINConv(a, b, agg) for a, b in zip(dims[:-1], dims[1:])])

        def forward():
            h = X
            for i, conv in enumerate(layers):
                h = conv(h)
                if i + 1 < len(layers):
                    h = F.relu(h)
            return F.log_softmax(h, dim=1)

    optimizer = torch.optim.Adam(layers.parameters(), lr=0.01)

    def train():
        optimizer.zero_grad()
        logp = forward()
        # mean negative log-likelihood over all nodes of the graph: this rank's share
        loss = -logp.gather(1, y.view(-1, 1)).sum() / n_global
        loss.backward()
        optimizer.step()
        return loss

    def sync():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    for _ in range(10):   # dry run
        train()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.num_epoches):
        loss = train()
    sync()
    elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    total = loss.detach().clone()
    dist.all_reduce(total)
    w0 = layers[0].weights.detach()
    w_ref = w0.clone()
    dist.broadcast(w_ref, src=0)
    in_sync = torch.tensor([1.0 if torch.equal(w0, w_ref) else 0.0], device=dev)
    dist.all_reduce(in_sync, op=dist.ReduceOp.MIN)
    if rank == 0:
        if verbose:
            print("# final loss: {:.6f}".format(float(total)))
            print("# weights identical on all ranks: {}".format(bool(in_sync.item())))
        print("Time (ms): {:.3f}".format(float(elapsed) * 1e3 / args.num_epoches))
        print()
    dist.barrier()
    dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
