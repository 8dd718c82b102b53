"""Probe of the source-major ("push") aggregation (tools/ceiling/push_probe.hip; VERDICT r4 task 2): builds the packed
arrays of the form with torch on the device -- sets of destination rows, their distinct sources in blocks of 256, the
(slot, register) edge lists per (block, wavefront) --, runs the kernel, checks it (X = ones exact, randn against fp64 rows and
against the library), and times it against the library's own kernel on the same graph.

    python tools/ceiling/probe_push.py [config] [sets] [ld]      (build first: tools/ceiling/build.sh)
"""
import ctypes
import json
import os
import sys
import time

import torch

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
from gnnadvisor_osdi21_amd import _lib, graph  # noqa: E402

lib = ctypes.CDLL(os.path.join(here, os.environ.get("PUSH_SO", "libpush.so")))
lib.push_launch.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_int, ctypes.c_int]
R = lib.push_rows_per_wave()
WAVES = 16
BLOCK = lib.push_block_rows() if hasattr(lib, "push_block_rows") else 256
dev = torch.device("cuda:0")
cfg = sys.argv[1] if len(sys.argv) > 1 else "reddit-like"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ld = int(sys.argv[3]) if len(sys.argv) > 3 else 128
D = 64


def pack(g, S):
    """-> dict of device arrays + statistics."""
    N, nnz = g.num_nodes, g.nnz
    rp = g.row_pointers.long()
    deg = rp[1:] - rp[:-1]
    # sets: equal shares of the edges, cut at row starts
    targets = (torch.arange(S + 1, device=dev, dtype=torch.float64) * (nnz / S)).long()
    bounds = torch.searchsorted(rp, targets, right=False).clamp_(0, N)
    bounds[0], bounds[-1] = 0, N
    rows_per_set = bounds[1:] - bounds[:-1]
    cap = WAVES * R
    assert int(rows_per_set.max()) <= cap, (int(rows_per_set.max()), cap)
    set_of_row = torch.searchsorted(bounds, torch.arange(N, device=dev), right=True) - 1
    # (wavefront, register) of every row: rows of a set by descending degree, dealt to the wavefronts in snake order
    order = torch.sort(set_of_row * (int(deg.max()) + 1) + (int(deg.max()) - deg), stable=True).indices
    rank = torch.empty(N, dtype=torch.long, device=dev)
    rank[order] = torch.arange(N, device=dev) - bounds[set_of_row[order]]
    rnd, pos = rank // WAVES, rank % WAVES
    wave_of_row = torch.where(rnd % 2 == 0, pos, WAVES - 1 - pos)
    reg_of_row = rnd
    store_row = torch.full((S * WAVES * R,), -1, dtype=torch.int32, device=dev)
    store_row[(set_of_row * WAVES + wave_of_row) * R + reg_of_row] = torch.arange(N, device=dev, dtype=torch.int32)
    # distinct (set, source) pairs, ascending
    dst = torch.repeat_interleave(torch.arange(N, device=dev), deg)
    src = g.column_index.long()
    key = set_of_row[dst] * N + src
    skey, perm = torch.sort(key)
    uniq, inv, counts = torch.unique_consecutive(skey, return_inverse=True, return_counts=True)
    u_set = uniq // N
    per_set = torch.bincount(u_set, minlength=S)
    first = torch.cumsum(per_set, 0) - per_set
    u_rank = torch.arange(uniq.numel(), device=dev) - first[u_set]
    blocks_per_set = (per_set + BLOCK - 1) // BLOCK
    blk_off = torch.zeros(S + 1, dtype=torch.long, device=dev)
    blk_off[1:] = torch.cumsum(blocks_per_set, 0)
    nblocks = int(blk_off[-1])
    src_ids = torch.zeros((nblocks + 2) * BLOCK, dtype=torch.int32, device=dev)       # (+ 2 blocks the prefetch may run into)
    src_ids[blk_off[u_set] * BLOCK + u_rank] = (uniq % N).to(torch.int32)
    # edges in (set, source) order -> (global block, wavefront) groups
    e_rank = u_rank[inv]
    e_set = u_set[inv]
    e_dst = dst[perm]
    gb = blk_off[e_set] + e_rank // BLOCK
    slot = e_rank % BLOCK
    grp = gb * WAVES + wave_of_row[e_dst]
    g_sorted, p2 = torch.sort(grp, stable=True)
    cnt = torch.bincount(g_sorted, minlength=nblocks * WAVES)
    padded = (cnt + 7) // 8 * 8
    ent_off = torch.zeros(nblocks * WAVES + 1 + 2 * WAVES, dtype=torch.long, device=dev)
    ent_off[1:nblocks * WAVES + 1] = torch.cumsum(padded, 0)
    ent_off[nblocks * WAVES + 1:] = ent_off[nblocks * WAVES]       # (the kernels look one block past a set's last)
    gstart = torch.cumsum(cnt, 0) - cnt
    within = torch.arange(nnz, device=dev) - gstart[g_sorted]
    total = int(ent_off[-1])
    per_block_entries = padded.view(nblocks, WAVES).sum(1)
    ent = torch.full((total + 16,), R, dtype=torch.int16, device=dev)                 # dummy: slot 0, register R
    val = (slot[p2] << 8 | reg_of_row[e_dst[p2]]).to(torch.int32)
    val = torch.where(val >= 32768, val - 65536, val).to(torch.int16)                 # (16-bit pattern)
    ent[ent_off[g_sorted] + within] = val
    per_blk = cnt.view(nblocks, WAVES).float()
    stats = dict(sets=S, max_rows_per_set=int(rows_per_set.max()), distinct_per_edge=round(uniq.numel() / nnz, 4),
                 blocks=nblocks, entries_padded_over_edges=round(total / nnz, 4),
                 mean_edges_per_block_wave=round(float(per_blk.mean()), 2), max_entries_per_block=int(per_block_entries.max()),
                 mean_of_block_max_over_mean=round(float((per_blk.max(1).values / per_blk.mean(1).clamp_min(1e-9)).mean()), 3),
                 packed_MB=round((src_ids.numel() * 4 + ent.numel() * 2 + ent_off.numel() * 4 + store_row.numel() * 4) / 1e6, 1))
    return dict(src_ids=src_ids, blk_off=blk_off.to(torch.int32), ent_off=ent_off.to(torch.int32).contiguous(),
                entries=ent, store_row=store_row, stats=stats)


if os.environ.get("PUSH_PACK_SELFTEST"):
    # the packing alone, on the CPU: a python walk of the packed arrays must reproduce A @ X
    dev = torch.device("cpu")
    g = graph.powerlaw_graph(3000, 60000, 900, seed=3)
    S = 4
    P = pack(g, S)
    X = torch.randn(g.num_nodes, 4, dtype=torch.float64)
    Y = torch.zeros(g.num_nodes, 4, dtype=torch.float64)
    ent = P["entries"].to(torch.int32) & 0xffff
    for s_ in range(S):
        acc = torch.zeros(WAVES, R + 1, 4, dtype=torch.float64)
        for b in range(int(P["blk_off"][s_]), int(P["blk_off"][s_ + 1])):
            ids = P["src_ids"][b * BLOCK:(b + 1) * BLOCK].long()
            for w_ in range(WAVES):
                lo, hi = int(P["ent_off"][b * WAVES + w_]), int(P["ent_off"][b * WAVES + w_ + 1])
                assert lo % 8 == 0 and hi % 8 == 0
                e = ent[lo:hi]
                acc[w_].index_add_(0, (e & 0xff).long(), X[ids[(e >> 8).long()]])
        rows = P["store_row"][s_ * WAVES * R:(s_ + 1) * WAVES * R].view(WAVES, R).long()
        ok = rows >= 0
        Y[rows[ok]] = acc[:, :R][ok]
    rp, ci = g.row_pointers.long(), g.column_index.long()
    ref = torch.zeros_like(Y)
    ref.index_add_(0, torch.repeat_interleave(torch.arange(g.num_nodes), rp[1:] - rp[:-1]), X[ci])
    print("selftest max diff", float((Y - ref).abs().max()), P["stats"])
    assert torch.allclose(Y, ref, atol=1e-9)
    sys.exit(0)
g = graph.make_config_graph(cfg, device=dev)
N, nnz = g.num_nodes, g.nnz
t0 = time.time()
P = pack(g, S)
torch.cuda.synchronize()
print(json.dumps(dict(cfg=cfg, N=N, nnz=nnz, pack_seconds_torch=round(time.time() - t0, 2), **P["stats"])), flush=True)

Xc = torch.randn(N, D, device=dev, generator=torch.Generator(device=dev).manual_seed(1234))
if ld != D:
    X = _lib.empty_rows(N, D, ld, dev)
    X.copy_(Xc)
else:
    X = Xc
Y = torch.full((N, D), float("nan"), device=dev)


def push(Xin, Yout):
    rc = lib.push_launch(Xin.data_ptr(), Yout.data_ptr(), P["src_ids"].data_ptr(), P["blk_off"].data_ptr(), P["ent_off"].data_ptr(),
                         P["entries"].data_ptr(), P["store_row"].data_ptr(), Xin.stride(0), S)
    assert rc == 0, rc


# correctness: X = ones -> exact row nnz; randn against fp64 rows and the library
ones = torch.ones_like(X) if X.is_contiguous() else _lib.empty_rows(N, D, ld, dev).fill_(1.0)
push(ones, Y)
torch.cuda.synchronize()
deg = (g.row_pointers[1:] - g.row_pointers[:-1]).float()
exact = bool((Y == deg[:, None]).all())
push(X, Y)
torch.cuda.synchronize()
rows = torch.randint(0, N, (200,), generator=torch.Generator().manual_seed(3)).tolist() + [int(torch.argmax(deg))]
worst = 0.0
for i in rows:
    b, e = int(g.row_pointers[i]), int(g.row_pointers[i + 1])
    ref = Xc[g.column_index[b:e].long()].double().sum(0)
    worst = max(worst, float(((Y[i].double() - ref).abs() / ref.abs().clamp_min(1.0)).max()))
ps = 128
pp, p2n = _lib.build_part(ps, g.row_pointers.cpu())
ppd, p2nd = pp.to(dev), p2n.to(dev)
Yl = _lib.sag(Xc, g.row_pointers, g.column_index, g.degrees, ppd, p2nd, ps, 32, 4)
torch.cuda.synchronize()
vs_lib = float(((Y - Yl).abs() / Yl.abs().clamp_min(1.0)).max())
Y2 = torch.empty_like(Y)
push(X, Y2)
torch.cuda.synchronize()
print(json.dumps(dict(ones_exact=exact, max_err_over_abs_ref=worst, max_diff_vs_library=vs_lib, nan_left=int(torch.isnan(Y).sum()),
                      bit_reproducible=bool(torch.equal(Y, Y2)))), flush=True)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return round(a.elapsed_time(b) / n, 4)


out = torch.empty_like(Xc)
res = dict(push_ms=timed(lambda: push(X, Y)))
_lib.prepare_graph(g.column_index, ppd, p2nd, N, N, ps, [D])
res["library_prepared_ms"] = timed(lambda: _lib.agg_ld(0, X, g.column_index, ppd, p2nd, N, ps, out=out))
res["library_phases"] = _lib.last_num_phases()
res["push_G_edges_per_s"] = round(nnz / res["push_ms"] / 1e6, 2)
print(json.dumps(res), flush=True)

# timing-only variants of the kernel (push_probe.hip built with -DPUSH_ABLATE=..., WRONG results): which part costs what
for name in [v for v in os.environ.get("PUSH_VARIANTS", "").split(",") if v]:
    so = os.path.join(here, f"libpush_{name}.so")
    if not os.path.exists(so):
        continue
    vlib = ctypes.CDLL(so)
    vlib.push_launch.argtypes = lib.push_launch.argtypes

    def vpush():
        rc = vlib.push_launch(X.data_ptr(), Y.data_ptr(), P["src_ids"].data_ptr(), P["blk_off"].data_ptr(), P["ent_off"].data_ptr(),
                              P["entries"].data_ptr(), P["store_row"].data_ptr(), X.stride(0), S)
        assert rc == 0, rc
    print(json.dumps({"variant": name, "ms": timed(vpush)}), flush=True)
