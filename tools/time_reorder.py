"""Host-side timing of the community renumbering, stage by stage (GNNA_REORDER_DEBUG), on a scrambled graph with hidden
locality.  Usage: python tools/time_reorder.py reddit-like|products-like [scale] [cache_dir]
Prints the stage laps (stderr of the library), total seconds, avg edge span before/after and the share of edges within 4,096 ids."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GNNA_REORDER_DEBUG", "1")
from gnnadvisor_osdi21_amd import _lib, graph

name = sys.argv[1] if len(sys.argv) > 1 else "reddit-like"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cache = sys.argv[3] if len(sys.argv) > 3 else "/tmp/rg"
path = os.path.join(cache, f"{name}_{scale}.npz")
if os.path.exists(path):
    z = np.load(path); src, dst, n = z["src"], z["dst"], int(z["n"])
else:
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    g = graph.make_config_graph(name, device=dev, locality=0.9, scale=scale)
    n = g.num_nodes
    rows = torch.repeat_interleave(torch.arange(n, device=dev), (g.row_pointers[1:] - g.row_pointers[:-1]).long())
    perm = torch.randperm(n, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    src = perm[rows].to(torch.int32).cpu().numpy(); dst = perm[g.column_index.long()].to(torch.int32).cpu().numpy()
    os.makedirs(cache, exist_ok=True)
    np.savez(path, src=src, dst=dst, n=n)
print(f"{name} x{scale}: {n} nodes, {len(src)} edges, {_lib.host_threads()} host threads", flush=True)
how = os.environ.get("TIME_REORDER_PATH", "csr")      # "csr": what loader.rabbit_reorder() does; "edges": gnna_reorder_community_i32
if how == "csr":
    t0 = time.perf_counter()
    rp, ci = _lib.csr_from_edges(src, dst, n)
    print(f"CSR build {time.perf_counter() - t0:.2f} s ({ci.numel()} entries)", flush=True)
    t0 = time.perf_counter()
    nid = _lib.reorder_community_csr(rp, ci, n)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    rp2, ci2 = _lib.relabel_csr(rp, ci, nid, n)
    s32, d32 = torch.from_numpy(src.astype(np.int32)), torch.from_numpy(dst.astype(np.int32))
    t2 = time.perf_counter()
    _lib.relabel_edges_(s32, d32, nid, n)
    print(f"relabel CSR {t2 - t1:.2f} s (incl. nothing else), relabel edge list in place {time.perf_counter() - t2:.2f} s", flush=True)
    new_id = nid.numpy()
else:
    t0 = time.perf_counter()
    new_id = _lib.reorder_community(src, dst, n).numpy()
    dt = time.perf_counter() - t0
s2, d2 = new_id[src], new_id[dst]
near = float(np.mean(np.abs(s2.astype(np.int64) - d2.astype(np.int64)) <= 4096))
print(f"renumbering ({how} path, SIMD={os.environ.get('GNNA_REORDER_SIMD', 'auto')}) total {dt:.2f} s; span {_lib.edge_span(src, dst):.0f} -> {_lib.edge_span(s2, d2):.0f}; within 4096 ids: {near:.4f}", flush=True)
if len(sys.argv) > 4:
    np.save(sys.argv[4], new_id)
