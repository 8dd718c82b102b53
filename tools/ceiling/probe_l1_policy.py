"""Can the CU's vector L1 be kept for the hot rows of a slice?  The bare access stream of the headline (Reddit-like, D = 64,
16 slices, phase-major ids) with the K most gathered rows of every slice loaded normally and every other row loaded with a
cache policy that bypasses the L1 (`sc1`, `sc0 sc1`; `sc0` and plain as controls), hot ids arranged first inside every
group of `arrange` ids (a load instruction has ONE policy).  tools/ceiling/l1_policy_probe.hip.
usage: probe_l1_policy.py [config] [phases]   (build first: tools/ceiling/build.sh)"""
import ctypes
import json
import os
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
from gnnadvisor_osdi21_amd import graph  # noqa: E402

lib = ctypes.CDLL(os.path.join(here, "libl1policy.so"))
lib.l1_policy_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
cfg = sys.argv[1] if len(sys.argv) > 1 else "reddit-like"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
D, SEG = 64, 512
g = graph.make_config_graph(cfg, device=dev)
N, nnz = g.num_nodes, g.column_index.numel()
col = g.column_index
out = torch.empty((nnz // 256 + 64) * 256, device=dev)
freq = torch.bincount(col.long(), minlength=N)
phase_of = torch.div(torch.arange(N, device=dev) * B, N, rounding_mode="floor")
order = torch.sort(phase_of[col.long()].to(torch.int16), stable=True).indices
base_ids = col[order].contiguous()
del order
print(json.dumps(dict(cfg=cfg, N=N, nnz=nnz, phases=B, seg=SEG)), flush=True)


def ms(X, ids, row_bytes, aux, n=10):
    def go():
        rc = lib.l1_policy_launch(X.data_ptr(), ids.data_ptr(), ids.numel(), SEG, row_bytes, aux, out.data_ptr())
        assert rc == 0, rc
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        go()
    b.record()
    torch.cuda.synchronize()
    return round(a.elapsed_time(b) / n, 4)


def arranged(K, arrange):
    """ids with the K most gathered rows of every slice unmarked (hot) and first inside every group of `arrange` ids, every
    other id marked cold (top bit) -> (ids, share of the edges that are hot)"""
    hot = torch.zeros(N, dtype=torch.bool, device=dev)
    if K > 0:
        for p in range(B):
            idx = (phase_of == p).nonzero().flatten()
            hot[idx[torch.topk(freq[idx], min(K, idx.numel())).indices]] = True
    is_cold = ~hot[base_ids.long()]
    key = (torch.arange(nnz, device=dev) // arrange) * 2 + is_cold.long()
    perm = torch.sort(key, stable=True).indices
    ids = base_ids[perm].long()
    cold = is_cold[perm]
    marked = torch.where(cold, (ids | 0x80000000) - (1 << 32), ids).to(torch.int32)
    return marked.contiguous(), round(float((~is_cold).float().mean()), 4)


for row_bytes in (512, 256):
    ld = row_bytes // 4
    X = torch.randn(N, ld, device=dev)
    rec = dict(row_bytes=row_bytes, plain_edge_order=ms(X, base_ids, row_bytes, 0))
    # everything cold: what does bypassing the L1 cost / gain by itself?
    allcold, _ = arranged(0, SEG)
    rec["all_rows_bypass"] = {name: ms(X, allcold, row_bytes, aux) for name, aux in (("sc0", 1), ("sc1", 16), ("sc0_sc1", 17))}
    del allcold
    print(json.dumps(rec), flush=True)
    for arrange in (512, 64):
        for K in (32, 64, 128, 256):
            ids, share = arranged(K, arrange)
            r = dict(row_bytes=row_bytes, hot_rows_per_slice=K, hot_first_within=arrange, hot_edge_share=share,
                     ms={name: ms(X, ids, row_bytes, aux) for name, aux in (("cold_plain", 0), ("cold_sc0", 1), ("cold_sc1", 16), ("cold_sc0_sc1", 17))})
            print(json.dumps(r), flush=True)
            del ids
    del X
