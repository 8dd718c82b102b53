"""The native community renumbering (gnna_reorder_community_i32, the product) beside a restatement of Rabbit Order (the
reference's renumbering, oracle/rabbit_yardstick.cpp -- the yardstick) on the same graphs with scrambled ids: the
reference's locality measure (mean |src - dst|, dataset.py:99-100), the share of edges within 4,096 / 16,384 ids, host
seconds, and -- when a GPU is present -- the aggregation time of the library on each order (reference caller's path,
D = 64).  CPU part runs anywhere.
usage: probe_reorder_quality.py [config[,config..]] [scale] [locality]      (defaults: reddit-like 1.0 0.9)"""
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
gpu = torch.cuda.is_available()
dev = torch.device("cuda:0" if gpu else "cpu")


def spans(src, dst):
    d = (src.long() - dst.long()).abs()
    return dict(avg_edge_span=round(float(d.double().mean()), 1), within_4096=round(float((d <= 4096).double().mean()), 4),
                within_16384=round(float((d <= 16384).double().mean()), 4))


def agg_ms(src, dst, n):
    """the library's own schedule through the reference caller's path (module, contiguous X, fresh output)"""
    from gnnadvisor_osdi21_amd import load_extension
    GNNA = load_extension()
    g = graph.graph_from_edges(src.to(dev), dst.to(dev), n)
    ps = 128 if g.avg_degree >= 256 else 32
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
    g = graph.make_config_graph(cfg, device="cpu", locality=locality, scale=scale)
    n = g.num_nodes
    rows = torch.repeat_interleave(torch.arange(n), (g.row_pointers[1:] - g.row_pointers[:-1]).long())
    cols = g.column_index.long()
    print(json.dumps(dict(config=cfg, scale=scale, hidden_locality=locality, num_nodes=n, nnz=int(cols.numel()))), flush=True)
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(1))
    src, dst = perm[rows].to(torch.int32), perm[cols].to(torch.int32)
    orders = {"planted (generator's)": (rows.to(torch.int32), cols.to(torch.int32), None), "scrambled": (src, dst, None)}
    t0 = time.perf_counter()
    ours = _lib.reorder_community(src, dst, n).long()
    orders["gnna_reorder_community_i32 (product)"] = (ours[src.long()].to(torch.int32), ours[dst.long()].to(torch.int32),
                                                      dict(seconds=round(time.perf_counter() - t0, 1), threads=os.cpu_count()))
    t0 = time.perf_counter()
    rb, st = oracle.rabbit_yardstick(src.numpy(), dst.numpy(), n)
    rb = torch.from_numpy(rb).long()
    assert np.array_equal(np.sort(rb.numpy()), np.arange(n))
    orders["Rabbit Order restatement (yardstick, 1 thread)"] = (
        rb[src.long()].to(torch.int32), rb[dst.long()].to(torch.int32),
        dict(seconds=round(time.perf_counter() - t0, 1), communities=st["communities"], modularity=round(st["modularity"], 4),
             aggregation_seconds=round(st["aggregation_seconds"], 1)))
    for name, (s, d, extra) in orders.items():
        rec = dict(order=name, **spans(s, d), **(extra or {}))
        if gpu:
            rec["sag_ms_D64"] = agg_ms(s, d, n)
        print(json.dumps(rec), flush=True)
