"""SDDMM on the headline graph: the floor of its access stream against the product (round 6, VERDICT r5 task 7).

  python tools/ceiling/probe_sddmm_floor.py [config] [dim]

Prints JSON lines: the bare gather stream of the schedule's slices (tools/ceiling/gather_ceiling.hip: gather_ceiling), the same
stream with the destination piece in registers, a dot product per gathered row and one coalesced 4-byte store per edge
(sddmm_ceiling = the floor of edge_out[e] = <A[row(e)], X[col(e)]>), and the product's gnna_sddmm_ld_f32 under its knobs."""
import ctypes, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gnnadvisor_osdi21_amd import _lib, graph  # noqa: E402

dev = torch.device("cuda:0")
cfg = sys.argv[1] if len(sys.argv) > 1 else "reddit-like"
D = int(sys.argv[2]) if len(sys.argv) > 2 else 64
g = graph.make_config_graph(cfg, device=dev)
N, E = g.num_nodes, g.nnz
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "ceiling", "libceiling.so"))
lib.gather_ceiling_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
lib.sddmm_ceiling_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                     ctypes.c_int64, ctypes.c_void_p]


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


A = torch.randn(N, D, device=dev)
X = torch.randn(N, D, device=dev)
out = torch.empty(E + 1024, device=dev)

# ---- the product, knob by knob
results = []
for ps in (32, 64, 128):
    pp, p2n = _lib.build_part(ps, g.row_pointers.cpu())
    ppd, p2nd = pp.to(dev), p2n.to(dev)
    for tune, ld in (({}, D), (dict(loads_in_flight=8), D), ({}, 128), (dict(loads_in_flight=8), 128), (dict(column_phases=1), D),
                     (dict(column_phases=4), D), (dict(column_phases=16), 128)):
        if ld != D and D != 64:
            continue
        _lib.reset_tuning(); _lib.set_tuning(**tune)
        Xl = _lib.empty_rows(N, D, ld, dev); Xl.copy_(X)
        ms = timed(lambda: _lib.sddmm(A, Xl, g.column_index, ppd, p2nd, ps, out=out[:E]))
        rec = dict(what="product gnna_sddmm_ld_f32", partSize=ps, tune=tune, ld_src=ld, ms=round(ms, 4), phases=_lib.last_num_phases(),
                   G_edges_s=round(E / ms / 1e6, 1))
        results.append(rec); print(json.dumps(rec), flush=True)
_lib.reset_tuning()
best = min(results, key=lambda r: r["ms"])

# ---- the floors, on the slices of the product's best schedule and on a few others
if D == 64:
    Xg = torch.zeros(2 * N, 64, device=dev); Xg[0::2] = X         # rows on 512-byte boundaries (the library's staged layout)
    for B in sorted({1, 4, 8, 16, int(best["phases"])}):
        col = g.column_index
        if B > 1:
            slice_rows = (N + 31) // 32
            ph = torch.div((col // slice_rows) * B, 32, rounding_mode="floor").to(torch.int16)
            ids = col[torch.sort(ph, stable=True).indices].contiguous()
        else:
            ids = col
        ids2 = (ids * 2).contiguous()
        small = torch.empty((E // 256 + 64) * 256, device=dev)
        for seg, U in ((512, 8), (512, 4), (256, 8)):
            g_ms = timed(lambda: lib.gather_ceiling_launch(Xg.data_ptr(), ids2.data_ptr(), ids2.numel(), 64, seg, U, small.data_ptr()))
            s_ms = timed(lambda: lib.sddmm_ceiling_launch(Xg.data_ptr(), ids2.data_ptr(), ids2.numel(), 64, seg, U, A.data_ptr(), N, out.data_ptr()))
            print(json.dumps(dict(what="floor", phases=B, seg=seg, U=U, gather_stream_ms=round(g_ms, 4), sddmm_floor_ms=round(s_ms, 4),
                                  store_and_dot_cost_ms=round(s_ms - g_ms, 4))), flush=True)
print(json.dumps(dict(what="summary", best_product_ms=best["ms"], best_product=best)), flush=True)
