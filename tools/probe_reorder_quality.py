"""The native community renumbering (gnna_reorder_community_i32, the product) beside a restatement of Rabbit Order (the
reference's renumbering, oracle/rabbit_yardstick.cpp -- the yardstick) on the same graphs with scrambled ids: the
reference's locality measure (mean |src - dst|, dataset.py:99-100), the share of edges within 4,096 / 16,384 ids, host
seconds, and -- when a GPU is present -- the aggregation time of the library on each order (reference caller's path,
D = 64).  CPU part runs anywhere.
usage: probe_reorder_quality.py [config[,config..]] [scale] [locality]      (defaults: reddit-like 1.0 0.9)
a config is a BASELINE stand-in (reddit-like, products-like: hidden locality = edges within +-4,096 ids of a planted order)
or  blocks:<nodes>:<entries>:<communities>  (degree-corrected block model, 90 % of the edges inside planted communities --
the structure Rabbit Order is made for)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402  (measurement tool: the yardstick lives with the test infrastructure)
from gnnadvisor_osdi21_amd import _lib, graph  # noqa: E402

configs = (sys.argv[1] if len(sys.argv) > 1 else "reddit-like").split(",")
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
locality = float(sys.argv[3]) if len(sys.argv) > 3 else 0.9
D = 64
# permutations computed once (e.g. on a CPU-only machine) travel in this directory, keyed by graph and a checksum of its edge
# list, so that a GPU box only times the aggregation (git-ignored: tools/_perm/)
PERM_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_perm")
gpu = torch.cuda.is_available()
dev = torch.device("cuda:0" if gpu else "cpu")


def spans(src, dst):
    d = (src.long() - dst.long()).abs()
    return dict(avg_edge_span=round(float(d.double().mean()), 1), within_4096=round(float((d <= 4096).double().mean()), 4),
                within_16384=round(float((d <= 16384).double().mean()), 4))


def cached(tag, src, dst, n, compute):
    """(new_id int64 tensor, extra dict) -- from PERM_DIR when the edge list is the one the file was made from"""
    chk = int(((src.long() * 1000003 + dst.long()) % 2147483629).sum())
    path = os.path.join(PERM_DIR, tag + ".npz")
    if os.path.exists(path):
        z = np.load(path, allow_pickle=False)
        if int(z["checksum"]) == chk and int(z["n"]) == n:
            extra = json.loads(str(z["extra"]))
            extra["precomputed"] = True
            return torch.from_numpy(z["new_id"]).long(), extra
    new_id, extra = compute()
    os.makedirs(PERM_DIR, exist_ok=True)
    np.savez(path, new_id=new_id.numpy().astype(np.int32), checksum=np.int64(chk), n=np.int64(n), extra=json.dumps(extra))
    return new_id, extra


def agg_ms(src, dst, n):
    """the library's own schedule through the reference caller's path (module, contiguous X, fresh output)"""
    from gnnadvisor_osdi21_amd import load_extension
    GNNA = load_extension()
    g = graph.graph_from_edges(src.to(dev), dst.to(dev), n)
    from gnnadvisor_osdi21_amd.decider import inputProperty

    class _Profile:
        pass
    prof = _Profile()
    prof.num_nodes, prof.avg_degree, prof.avg_edgeSpan = g.num_nodes, g.avg_degree, g.avg_edgeSpan
    prof.num_features, prof.reorder_flag = D, False
    prof.rabbit_reorder = lambda: None
    info = inputProperty(None, None, None, 32, 32, 4, 100, hiddenDim=D, dataset_obj=prof, enable_rabbit=False, manual_mode=False)
    info.decider()
    ps = info.partSize
    pp, p2n = GNNA.build_part(ps, g.row_pointers.cpu())
    ppd, p2nd = pp.to(dev), p2n.to(dev)
    X = torch.randn(n, D, device=dev)
    fn = lambda: GNNA.SAG(X, g.row_pointers, g.column_index, g.degrees, ppd, p2nd, ps, 32, 4)  # noqa: E731
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        fn()
    b.record()
    torch.cuda.synchronize()
    return round(a.elapsed_time(b) / 20, 4)


for cfg in configs:
    if cfg.startswith("blocks:"):
        bn, be, bk = (int(v) for v in cfg.split(":")[1:4])
        g = graph.community_graph(int(bn * scale), int(be * scale), bk, p_in=locality, seed=11)
    else:
        g = graph.make_config_graph(cfg, device="cpu", locality=locality, scale=scale)
    n = g.num_nodes
    rows = torch.repeat_interleave(torch.arange(n), (g.row_pointers[1:] - g.row_pointers[:-1]).long())
    cols = g.column_index.long()
    print(json.dumps(dict(config=cfg, scale=scale, hidden_locality=locality, num_nodes=n, nnz=int(cols.numel()))), flush=True)
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(1))
    src, dst = perm[rows].to(torch.int32), perm[cols].to(torch.int32)
    orders = {"planted (generator's)": (rows.to(torch.int32), cols.to(torch.int32), None), "scrambled": (src, dst, None)}
    key = f"{cfg.replace(':', '-')}_scale{scale}_loc{locality}"

    def run_ours():
        t0 = time.perf_counter()
        p = _lib.reorder_community(src, dst, n).long()
        return p, dict(seconds=round(time.perf_counter() - t0, 1), threads=os.cpu_count())

    def run_rabbit():
        t0 = time.perf_counter()
        p, st = oracle.rabbit_yardstick(src.numpy(), dst.numpy(), n)
        return torch.from_numpy(p).long(), dict(seconds=round(time.perf_counter() - t0, 1), communities=st["communities"],
                                                modularity=round(st["modularity"], 4),
                                                aggregation_seconds=round(st["aggregation_seconds"], 1))

    ours, ex = cached(key + "_product", src, dst, n, run_ours)
    assert np.array_equal(np.sort(ours.numpy()), np.arange(n))
    orders["gnna_reorder_community_i32 (product)"] = (ours[src.long()].to(torch.int32), ours[dst.long()].to(torch.int32), ex)
    rb, ex = cached(key + "_rabbit", src, dst, n, run_rabbit)
    assert np.array_equal(np.sort(rb.numpy()), np.arange(n))
    orders["Rabbit Order restatement (yardstick, 1 thread)"] = (rb[src.long()].to(torch.int32), rb[dst.long()].to(torch.int32), ex)
    for name, (s, d, extra) in orders.items():
        rec = dict(order=name, **spans(s, d), **(extra or {}))
        if gpu:
            rec["sag_ms_D64"] = agg_ms(s, d, n)
        print(json.dumps(rec), flush=True)
