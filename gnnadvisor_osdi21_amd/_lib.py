"""ctypes binding of the C ABI in ``include/gnna.h`` (``csrc/libgnna.so``).

PyTorch is used here only as the owner of device memory and streams: every call hands
raw ``data_ptr()`` addresses and the current HIP stream handle to the library.  There
is no CPU fallback -- if the shared library is missing the import of the product path
fails loudly.
"""
from __future__ import annotations

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GNNA_LIB") or os.path.join(_HERE, "csrc", "libgnna.so")  # GNNA_LIB: A/B builds

GNNA_OK = 0


class GnnaError(RuntimeError):
    pass


class Tuning(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_int), ("groups_per_chunk", ctypes.c_int), ("loads_in_flight", ctypes.c_int),
                ("blocks_per_cu", ctypes.c_int), ("xcd_remap", ctypes.c_int),
                ("trust_canonical", ctypes.c_int), ("column_phases", ctypes.c_int),
                ("avg_degree", ctypes.c_int), ("nonlocal_ids", ctypes.c_int), ("gcn_prescale", ctypes.c_int),
                ("pad_rows", ctypes.c_int), ("zero_fill", ctypes.c_int),
                ("sweep", ctypes.c_int), ("sweep_slack", ctypes.c_int), ("deterministic", ctypes.c_int),
                ("pack_ids", ctypes.c_int), ("ids_check_every", ctypes.c_int), ("wide_blocks", ctypes.c_int)]


_lib = None

_AGG_COMMON = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]  # input, row_pointers, column_index
_TAIL = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,        # part_pointers, part2Node, out
         ctypes.c_int64, ctypes.c_int, ctypes.c_int64,             # num_nodes, dim, num_parts
         ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]  # partSize, dimWorker, warpPerBlock, stream

EXPORTS = ("gnna_version", "gnna_build_id", "gnna_last_error", "gnna_count_parts", "gnna_build_part_i32",
           "gnna_sag_f32", "gnna_agg_gcn_f32", "gnna_agg_gin_f32", "gnna_set_tuning", "gnna_get_tuning",
           "gnna_profile_begin", "gnna_profile_end", "gnna_agg_rect_f32",
           "gnna_csr_from_edges_i32", "gnna_degrees_f32", "gnna_edge_span", "gnna_reorder_rcm_i32",
           "gnna_last_num_phases", "gnna_sddmm_f32", "gnna_sddmm_ld_f32", "gnna_agg_rect_windows_f32", "gnna_set_graph_hints", "gnna_xtg_f32", "gnna_set_graph_phases",
           "gnna_last_num_launches", "gnna_reorder_community_i32", "gnna_prepare_graph", "gnna_release_graph",
           "gnna_runtime_counters", "gnna_row_counts_i64", "gnna_row_splits_i64", "gnna_csr_from_edges_range_i32",
           "gnna_forget_graph", "gnna_agg_ld_f32", "gnna_preferred_ld", "gnna_device_cus", "gnna_host_threads",
           "gnna_reorder_community_csr_i32", "gnna_relabel_edges_i32", "gnna_relabel_csr_i32", "gnna_runtime_counters_ex", "gnna_forget_plans")


def load() -> ctypes.CDLL:
    """dlopen libgnna.so (built by ``python -m gnnadvisor_osdi21_amd.build``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python -m gnnadvisor_osdi21_amd.build`). There is no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    L.gnna_version.restype = ctypes.c_int
    L.gnna_build_id.restype = ctypes.c_char_p
    L.gnna_last_error.restype = ctypes.c_char_p
    L.gnna_device_cus.restype = ctypes.c_int
    L.gnna_host_threads.restype = ctypes.c_int
    L.gnna_count_parts.restype = ctypes.c_int64
    L.gnna_count_parts.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64]
    L.gnna_build_part_i32.restype = ctypes.c_int
    L.gnna_build_part_i32.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    L.gnna_sag_f32.restype = ctypes.c_int
    L.gnna_sag_f32.argtypes = _AGG_COMMON + [ctypes.c_void_p] + _TAIL
    L.gnna_agg_gcn_f32.restype = ctypes.c_int
    L.gnna_agg_gcn_f32.argtypes = _AGG_COMMON + [ctypes.c_void_p] + _TAIL
    L.gnna_agg_gin_f32.restype = ctypes.c_int
    L.gnna_agg_gin_f32.argtypes = _AGG_COMMON + [ctypes.c_float] + _TAIL
    L.gnna_set_tuning.restype = ctypes.c_int
    L.gnna_set_tuning.argtypes = [ctypes.POINTER(Tuning)]
    L.gnna_get_tuning.restype = None
    L.gnna_get_tuning.argtypes = [ctypes.POINTER(Tuning)]
    L.gnna_agg_rect_f32.restype = ctypes.c_int
    L.gnna_agg_rect_f32.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p,
                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                                    ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    L.gnna_agg_ld_f32.restype = ctypes.c_int
    L.gnna_agg_ld_f32.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
                                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p,
                                  ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int64,
                                  ctypes.c_int, ctypes.c_uint, ctypes.c_void_p]
    L.gnna_preferred_ld.restype = ctypes.c_int64
    L.gnna_preferred_ld.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int64]
    L.gnna_agg_rect_windows_f32.restype = ctypes.c_int
    L.gnna_agg_rect_windows_f32.argtypes = (L.gnna_agg_rect_f32.argtypes[:-1]
                                            + [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p])
    L.gnna_set_graph_hints.restype = ctypes.c_int
    L.gnna_set_graph_hints.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    L.gnna_set_graph_phases.restype = ctypes.c_int
    L.gnna_set_graph_phases.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    L.gnna_xtg_f32.restype = ctypes.c_int
    L.gnna_xtg_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                               ctypes.c_int, ctypes.c_void_p]
    L.gnna_csr_from_edges_i32.restype = ctypes.c_int64
    L.gnna_csr_from_edges_i32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                          ctypes.c_void_p, ctypes.c_void_p]
    L.gnna_row_counts_i64.restype = ctypes.c_int
    L.gnna_row_counts_i64.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
    L.gnna_row_splits_i64.restype = ctypes.c_int
    L.gnna_row_splits_i64.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    L.gnna_csr_from_edges_range_i32.restype = ctypes.c_int64
    L.gnna_csr_from_edges_range_i32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                                ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_int64]
    L.gnna_degrees_f32.restype = ctypes.c_int
    L.gnna_degrees_f32.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    L.gnna_edge_span.restype = ctypes.c_int
    L.gnna_edge_span.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_double)]
    L.gnna_reorder_rcm_i32.restype = ctypes.c_int
    L.gnna_reorder_rcm_i32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                       ctypes.c_void_p]
    L.gnna_reorder_community_i32.restype = ctypes.c_int
    L.gnna_reorder_community_i32.argtypes = L.gnna_reorder_rcm_i32.argtypes
    L.gnna_reorder_community_csr_i32.restype = ctypes.c_int
    L.gnna_reorder_community_csr_i32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    L.gnna_relabel_edges_i32.restype = ctypes.c_int
    L.gnna_relabel_edges_i32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                         ctypes.POINTER(ctypes.c_double)]
    L.gnna_relabel_csr_i32.restype = ctypes.c_int
    L.gnna_relabel_csr_i32.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int64] + [ctypes.c_void_p] * 3
    L.gnna_last_num_phases.restype = ctypes.c_int
    L.gnna_last_num_launches.restype = ctypes.c_int
    L.gnna_sddmm_ld_f32.restype = ctypes.c_int
    L.gnna_sddmm_ld_f32.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_void_p] * 4 + [
        ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
    L.gnna_sddmm_f32.restype = ctypes.c_int
    L.gnna_sddmm_f32.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int64,
                                                        ctypes.c_int, ctypes.c_void_p]
    L.gnna_prepare_graph.restype = ctypes.c_int
    L.gnna_prepare_graph.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                     ctypes.c_int64, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_int,
                                     ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]
    L.gnna_release_graph.restype = ctypes.c_int
    L.gnna_release_graph.argtypes = [ctypes.c_void_p]
    L.gnna_forget_graph.restype = ctypes.c_int
    L.gnna_forget_graph.argtypes = [ctypes.c_void_p]
    L.gnna_forget_plans.restype = ctypes.c_int
    L.gnna_forget_plans.argtypes = [ctypes.c_void_p]
    L.gnna_runtime_counters.restype = None
    L.gnna_runtime_counters.argtypes = [ctypes.POINTER(ctypes.c_int64)]
    L.gnna_runtime_counters_ex.restype = ctypes.c_int
    L.gnna_runtime_counters_ex.argtypes = [ctypes.POINTER(ctypes.c_int64), ctypes.c_int]
    L.gnna_profile_begin.restype = ctypes.c_int
    L.gnna_profile_begin.argtypes = [ctypes.c_int]
    L.gnna_profile_end.restype = ctypes.c_int
    L.gnna_profile_end.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                   ctypes.POINTER(ctypes.c_int)]
    _lib = L
    return L


def _check(rc: int) -> None:
    if rc != GNNA_OK:
        raise GnnaError(f"libgnna error {rc}: {load().gnna_last_error().decode()}")


def _ptr(t: torch.Tensor | None) -> int | None:
    return None if t is None else t.data_ptr()


def _stream(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def set_tuning(groups_per_chunk=-1, loads_in_flight=-1, blocks_per_cu=-1, xcd_remap=-1,
               trust_canonical=-1, column_phases=-1, avg_degree=-1, nonlocal_ids=-1, gcn_prescale=-1,
               pad_rows=-1, zero_fill=-1, sweep=-1, sweep_slack=-1, deterministic=-1, pack_ids=-1,
               wide_blocks=-1, ids_check_every=-1) -> None:
    """Fields < 0 (<= 0 where 0 has no meaning) keep the built-in choice; ids_check_every: 1 = a full hash of the ids
    behind a packed copy on every call, n = every n-th (built-in 64), >= 2**30 = never (include/gnna.h)."""
    t = Tuning(ctypes.sizeof(Tuning), groups_per_chunk, loads_in_flight, blocks_per_cu, xcd_remap, trust_canonical, column_phases,
               avg_degree, nonlocal_ids, gcn_prescale, pad_rows, zero_fill, sweep, sweep_slack, deterministic,
               pack_ids, ids_check_every, wide_blocks)
    _check(load().gnna_set_tuning(ctypes.byref(t)))


def reset_tuning() -> None:
    _check(load().gnna_set_tuning(None))


def device_cus() -> int:
    """Compute units of the current device (0: no device visible)."""
    return int(load().gnna_device_cus())


def host_threads() -> int:
    """Host threads the native builders and the renumbering use (affinity, cgroup quota, GNNA_HOST_THREADS)."""
    return int(load().gnna_host_threads())


def build_id() -> str:
    """"0.6.0+<source hash>" of the loaded libgnna.so (gnna_build_id)."""
    return load().gnna_build_id().decode()


_registered: dict = {}      # device address -> token of the storage that registered hints / schedules for it


def _forget_when_freed(column_index) -> None:
    """libgnna keys its per-graph tables by the device address of `column_index`; PyTorch's caching allocator
    hands that address to an unrelated tensor once this one is freed.  Drop the entries together with the
    memory.  The finalizer hangs on the tensor's STORAGE, not on the Python tensor object: hints registered
    through a temporary alias (a view, a slice, `.detach()`) must live as long as the graph does, not as long as
    the alias (unless a newer storage has registered the same address in the meantime)."""
    import weakref
    ptr = column_index.data_ptr()
    storage = column_index.untyped_storage()
    token = _registered.get(ptr)
    if token is not None and token[0]() is storage:
        return                                           # this storage already carries the finalizer for `ptr`
    token = (weakref.ref(storage),)
    _registered[ptr] = token

    def drop(ptr=ptr, token=token):
        if _registered.get(ptr) is token:
            del _registered[ptr]
            if _lib is not None:
                # (a finalizer runs whenever and wherever the garbage collector does: unlink only, free later --
                # gnna_release_graph would hipFree here, a device-wide synchronisation that breaks a capture in progress)
                _lib.gnna_forget_graph(ptr)
    weakref.finalize(storage, drop)


def set_graph_hints(column_index, avg_degree: float, nonlocal_ids: bool) -> None:
    """Per-graph hints (keyed by the device address of `column_index`): average edges per destination
    row and whether the source ids of a row are scattered over the whole id range.  avg_degree <= 0
    forgets the graph; column_index None forgets all.  The entry is dropped when the tensor is freed."""
    ptr = None if column_index is None else column_index.data_ptr()
    _check(load().gnna_set_graph_hints(ptr, int(avg_degree), 1 if nonlocal_ids else 0))
    if column_index is None:
        _registered.clear()
    elif avg_degree > 0:
        _forget_when_freed(column_index)


def get_tuning() -> dict:
    t = Tuning()
    load().gnna_get_tuning(ctypes.byref(t))
    return {name: getattr(t, name) for name, _ in Tuning._fields_ if name != "struct_size"}


def _host_i32(t) -> torch.Tensor:
    t = torch.as_tensor(t)
    return t.to(dtype=torch.int32, device="cpu").contiguous()


def csr_from_edges(src, dst, num_nodes: int):
    """Host CSR builder of the C ABI: (row_pointers int32 [N+1], column_index int32 [nnz])."""
    s, d = _host_i32(src), _host_i32(dst)
    assert s.numel() == d.numel()
    rp = torch.empty(int(num_nodes) + 1, dtype=torch.int32)
    ci = torch.empty(max(1, s.numel()), dtype=torch.int32)
    nnz = load().gnna_csr_from_edges_i32(s.data_ptr(), d.data_ptr(), s.numel(), int(num_nodes),
                                         rp.data_ptr(), ci.data_ptr())
    if nnz < 0:
        _check(int(nnz))
    return rp, ci[:nnz].clone()


def row_counts(rows, num_nodes: int, counts: torch.Tensor | None = None) -> torch.Tensor:
    """int64 [num_nodes]: list entries per row, accumulated into `counts` when given (feed a long list in pieces)."""
    r = _host_i32(rows)
    if counts is None:
        counts = torch.zeros(int(num_nodes), dtype=torch.int64)
    assert counts.dtype == torch.int64 and counts.numel() == int(num_nodes) and counts.is_contiguous()
    _check(load().gnna_row_counts_i64(r.data_ptr(), r.numel(), int(num_nodes), counts.data_ptr()))
    return counts


def row_splits(counts: torch.Tensor, world: int, want_row_pointers: bool = False):
    """-> (bounds: list of world + 1 row boundaries of nnz-balanced contiguous blocks[, global int64 row_pointers])."""
    assert counts.dtype == torch.int64 and counts.is_contiguous() and not counts.is_cuda
    n = counts.numel()
    bounds = torch.empty(int(world) + 1, dtype=torch.int64)
    rp = torch.empty(n + 1, dtype=torch.int64) if want_row_pointers else None
    _check(load().gnna_row_splits_i64(counts.data_ptr(), n, int(world), bounds.data_ptr(), _ptr(rp)))
    return (bounds.tolist(), rp) if want_row_pointers else bounds.tolist()


def csr_from_edges_range(src, dst, num_nodes: int, row_lo: int, row_hi: int, capacity: int | None = None):
    """The CSR rows [row_lo, row_hi) of an edge list: (local int32 row_pointers rebased to 0, GLOBAL int32 column ids)."""
    s, d = _host_i32(src), _host_i32(dst)
    assert s.numel() == d.numel()
    if capacity is None:
        capacity = int(((s >= row_lo) & (s < row_hi)).sum()) if s.numel() else 0
    rp = torch.empty(int(row_hi - row_lo) + 1, dtype=torch.int32)
    ci = torch.empty(max(1, int(capacity)), dtype=torch.int32)
    nnz = load().gnna_csr_from_edges_range_i32(s.data_ptr(), d.data_ptr(), s.numel(), int(num_nodes), int(row_lo),
                                               int(row_hi), rp.data_ptr(), ci.data_ptr(), int(capacity))
    if nnz < 0:
        _check(int(nnz))
    return rp, ci[:nnz].clone()


def degrees(row_pointers: torch.Tensor) -> torch.Tensor:
    rp = _host_i32(row_pointers)
    out = torch.empty(rp.numel() - 1, dtype=torch.float32)
    _check(load().gnna_degrees_f32(rp.data_ptr(), rp.numel() - 1, out.data_ptr()))
    return out


def edge_span(src, dst) -> float:
    s, d = _host_i32(src), _host_i32(dst)
    v = ctypes.c_double()
    _check(load().gnna_edge_span(s.data_ptr(), d.data_ptr(), s.numel(), ctypes.byref(v)))
    return v.value


def reorder_rcm(src, dst, num_nodes: int) -> torch.Tensor:
    """new_id[old_id] from the native reverse Cuthill-McKee renumbering."""
    s, d = _host_i32(src), _host_i32(dst)
    out = torch.empty(int(num_nodes), dtype=torch.int32)
    _check(load().gnna_reorder_rcm_i32(s.data_ptr(), d.data_ptr(), s.numel(), int(num_nodes), out.data_ptr()))
    return out


def reorder_community(src, dst, num_nodes: int) -> torch.Tensor:
    """new_id[old_id] from the native community renumbering (label propagation + chain + barycentre sweeps)."""
    s, d = _host_i32(src), _host_i32(dst)
    out = torch.empty(int(num_nodes), dtype=torch.int32)
    _check(load().gnna_reorder_community_i32(s.data_ptr(), d.data_ptr(), s.numel(), int(num_nodes), out.data_ptr()))
    return out


def reorder_community_csr(row_pointers, column_index, num_nodes: int) -> torch.Tensor:
    """new_id[old_id] of the community renumbering from a host CSR with sorted, duplicate-free rows (the loader's): a symmetric
    CSR is the algorithm's adjacency as it stands (no second counting sort); same permutation as `reorder_community`."""
    rp, ci = _host_i32(row_pointers), _host_i32(column_index)
    assert rp.numel() == int(num_nodes) + 1
    out = torch.empty(int(num_nodes), dtype=torch.int32)
    _check(load().gnna_reorder_community_csr_i32(rp.data_ptr(), ci.data_ptr(), int(num_nodes), out.data_ptr()))
    return out


def relabel_edges_(src: torch.Tensor, dst: torch.Tensor, new_id: torch.Tensor, num_nodes: int) -> float:
    """src, dst (int32 host tensors, contiguous) <- new_id[...] IN PLACE; returns the new mean |src - dst|."""
    for t in (src, dst, new_id):
        assert t.dtype == torch.int32 and t.is_contiguous() and not t.is_cuda
    v = ctypes.c_double()
    _check(load().gnna_relabel_edges_i32(src.data_ptr(), dst.data_ptr(), src.numel(), new_id.data_ptr(), int(num_nodes), ctypes.byref(v)))
    return v.value


def relabel_csr(row_pointers, column_index, new_id, num_nodes: int):
    """The relabelled graph's CSR (rows sorted) from the old CSR and new_id[old] -- no global sort."""
    rp, ci, nid = _host_i32(row_pointers), _host_i32(column_index), _host_i32(new_id)
    out_rp = torch.empty(int(num_nodes) + 1, dtype=torch.int32)
    out_ci = torch.empty(max(1, ci.numel()), dtype=torch.int32)
    _check(load().gnna_relabel_csr_i32(rp.data_ptr(), ci.data_ptr(), int(num_nodes), nid.data_ptr(), out_rp.data_ptr(), out_ci.data_ptr()))
    return out_rp, out_ci[:ci.numel()]


def last_num_phases() -> int:
    return int(load().gnna_last_num_phases())


def last_num_launches() -> int:
    return int(load().gnna_last_num_launches())


def profile_begin(max_calls: int) -> None:
    _check(load().gnna_profile_begin(int(max_calls)))


def profile_end() -> dict:
    """-> {main_ms, prologue_ms, calls}: HIP-event averages of the two kernels per call."""
    a, b, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
    _check(load().gnna_profile_end(ctypes.byref(a), ctypes.byref(b), ctypes.byref(n)))
    return dict(main_ms=a.value, prologue_ms=b.value, calls=n.value)


def count_parts(partSize: int, indptr: torch.Tensor) -> int:
    assert indptr.dtype == torch.int32 and not indptr.is_cuda and indptr.is_contiguous()
    n = load().gnna_count_parts(int(partSize), indptr.data_ptr(), indptr.numel() - 1)
    if n < 0:
        _check(int(n))
    return int(n)


def build_part(partSize: int, indptr: torch.Tensor):
    """C-ABI partitioner -> (partPtr int32 [P+1], part2Node int32 [P]) on the CPU."""
    indptr = indptr.contiguous()
    P = count_parts(partSize, indptr)
    pp = torch.empty(P + 1, dtype=torch.int32)
    p2n = torch.empty(P, dtype=torch.int32)
    _check(load().gnna_build_part_i32(int(partSize), indptr.data_ptr(), indptr.numel() - 1,
                                      pp.data_ptr(), p2n.data_ptr(), P))
    return pp, p2n


def _fresh_output(shape, device) -> torch.Tensor:
    """A new output tensor: the library writes every element.  GNNA_DEBUG_POISON=1 (the test suite sets it) starts it
    as NaN, so that an element the library fails to write cannot hide behind whatever the allocator hands back."""
    if os.environ.get("GNNA_DEBUG_POISON", "0") not in ("", "0"):
        return torch.full(tuple(shape), float("nan"), dtype=torch.float32, device=device)
    return torch.empty(tuple(shape), dtype=torch.float32, device=device)


def _agg(fn, X, row_pointers, column_index, extra, part_pointers, part2Node, partSize, dimWorker,
         warpPerBlock, out):
    if not X.is_cuda:
        raise GnnaError("aggregation needs device tensors: there is no CPU path in libgnna")
    assert X.dtype == torch.float32 and X.is_contiguous() and X.dim() == 2
    for t in (column_index, part_pointers, part2Node):
        assert t.dtype == torch.int32 and t.is_contiguous() and t.device == X.device
    if out is None:
        out = _fresh_output(X.shape, X.device)
    with torch.cuda.device(X.device):
        _check(fn(X.data_ptr(), _ptr(row_pointers), column_index.data_ptr(), extra,
                  part_pointers.data_ptr(), part2Node.data_ptr(), out.data_ptr(),
                  X.shape[0], X.shape[1], part2Node.numel(),
                  int(partSize), int(dimWorker), int(warpPerBlock), _stream(X.device)))
    return out


def sag(X, row_pointers, column_index, degrees, part_pointers, part2Node, partSize=32, dimWorker=32,
        warpPerBlock=4, out=None):
    return _agg(load().gnna_sag_f32, X, row_pointers, column_index, _ptr(degrees), part_pointers,
                part2Node, partSize, dimWorker, warpPerBlock, out)


def agg_gcn(X, row_pointers, column_index, degrees, part_pointers, part2Node, partSize=32,
            dimWorker=32, warpPerBlock=4, out=None):
    assert degrees.dtype == torch.float32 and degrees.device == X.device
    return _agg(load().gnna_agg_gcn_f32, X, row_pointers, column_index, degrees.data_ptr(),
                part_pointers, part2Node, partSize, dimWorker, warpPerBlock, out)


def agg_gin(X, row_pointers, column_index, epsilon, part_pointers, part2Node, partSize=32,
            dimWorker=32, warpPerBlock=4, out=None):
    return _agg(load().gnna_agg_gin_f32, X, row_pointers, column_index, ctypes.c_float(epsilon),
                part_pointers, part2Node, partSize, dimWorker, warpPerBlock, out)


MODE_SAG, MODE_GCN, MODE_GIN = 0, 1, 2


def agg_rect(mode, X, column_index, part_pointers, part2Node, num_out_rows, partSize=32,
             degrees_out=None, degrees_in=None, epsilon=1.0, out=None, accumulate=False, windows=None):
    """Destination-shard aggregation: X is [num_in_rows, dim] (all sources), out is
    [num_out_rows, dim]; column_index indexes X.  ``windows=(K, begin, end)`` aggregates only the
    edges whose source lies in windows [begin, end) of K equal source windows
    (gnna_agg_rect_windows_f32: stateless; the column ids of every neighbor-group must be in increasing order -- the
    loader's CSR has them sorted -- else the call returns GNNA_ERR_UNSUPPORTED; the first call on a partition counts
    the ids per window and synchronises the stream once)."""
    if not X.is_cuda:
        raise GnnaError("aggregation needs device tensors: there is no CPU path in libgnna")
    assert X.dtype == torch.float32 and X.is_contiguous() and X.dim() == 2
    if out is None:
        assert not accumulate, "accumulate needs an existing `out`"
        out = _fresh_output((num_out_rows, X.shape[1]), X.device)
    if windows is not None:
        K, wb, we = (int(v) for v in windows)
        assert wb == 0 or out is not None
        with torch.cuda.device(X.device):
            _check(load().gnna_agg_rect_windows_f32(
                int(mode), X.data_ptr(), X.shape[0], column_index.data_ptr(), _ptr(degrees_out), _ptr(degrees_in),
                float(epsilon), part_pointers.data_ptr(), part2Node.data_ptr(), out.data_ptr(), int(num_out_rows),
                X.shape[1], part2Node.numel(), int(partSize), 1 if accumulate else 0, K, wb, we, _stream(X.device)))
        return out
    with torch.cuda.device(X.device):
        _check(load().gnna_agg_rect_f32(int(mode), X.data_ptr(), X.shape[0], column_index.data_ptr(),
                                        _ptr(degrees_out), _ptr(degrees_in), float(epsilon),
                                        part_pointers.data_ptr(), part2Node.data_ptr(), out.data_ptr(),
                                        int(num_out_rows), X.shape[1], part2Node.numel(), int(partSize),
                                        1 if accumulate else 0, _stream(X.device)))
    return out


ACCUMULATE, EPILOGUE_RELU = 1, 2


def _rows_view(t: torch.Tensor, what: str):
    """(data pointer, rows, dim, leading dimension) of a 2-D float32 device tensor whose rows are contiguous."""
    assert t.dtype == torch.float32 and t.dim() == 2, f"{what} must be a 2-D float32 tensor"
    if t.shape[1] > 1 and t.stride(1) != 1:
        raise GnnaError(f"{what}: the floats of a row must be contiguous (stride(1) == 1)")
    ld = t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))
    if ld < t.shape[1]:
        raise GnnaError(f"{what}: rows overlap (stride(0) = {t.stride(0)} < {t.shape[1]})")
    return t.data_ptr(), t.shape[0], t.shape[1], ld


def agg_ld(mode, X, column_index, part_pointers, part2Node, num_out_rows, partSize=32, degrees_out=None,
           degrees_in=None, epsilon=1.0, out=None, accumulate=False, relu=False):
    """gnna_agg_ld_f32: the rectangular aggregation with leading dimensions and the ReLU epilogue.  `X` and `out` may be
    row-strided views (a column block ``M[:, a:b]``, a padded buffer ``P[:, :dim]``): stride(1) must be 1, stride(0) is
    handed over as the leading dimension.  relu: out = max(out, 0) after the aggregation."""
    if not X.is_cuda:
        raise GnnaError("aggregation needs device tensors: there is no CPU path in libgnna")
    xp, n_in, dim, ld_in = _rows_view(X, "X")
    if out is None:
        assert not accumulate, "accumulate needs an existing `out`"
        out = _fresh_output((num_out_rows, dim), X.device)
    yp, n_out, dim_o, ld_out = _rows_view(out, "out")
    assert n_out == int(num_out_rows) and dim_o == dim and out.device == X.device
    flags = (ACCUMULATE if accumulate else 0) | (EPILOGUE_RELU if relu else 0)
    with torch.cuda.device(X.device):
        _check(load().gnna_agg_ld_f32(int(mode), xp, ld_in, n_in, column_index.data_ptr(), _ptr(degrees_out),
                                      _ptr(degrees_in), float(epsilon), part_pointers.data_ptr(), part2Node.data_ptr(),
                                      yp, ld_out, int(num_out_rows), dim, part2Node.numel(), int(partSize), flags,
                                      _stream(X.device)))
    return out


def preferred_ld(dim: int, num_in_rows: int, num_edges: int) -> int:
    """The leading dimension (floats) a producer should write `dim`-float source rows with so that the gather needs no
    staged copy (== dim: the contiguous layout is fine)."""
    return int(load().gnna_preferred_ld(int(dim), int(num_in_rows), int(num_edges)))


def empty_rows(num_rows: int, dim: int, ld: int, device) -> torch.Tensor:
    """A [num_rows, dim] float32 view with leading dimension `ld` of a fresh, (ld x 4)-byte aligned allocation."""
    if ld == dim:
        return torch.empty(num_rows, dim, dtype=torch.float32, device=device)
    buf = torch.empty(num_rows * ld + ld, dtype=torch.float32, device=device)
    off = (-buf.data_ptr() // 4) % ld
    return buf[off: off + num_rows * ld].view(num_rows, ld)[:, :dim]


def prepare_graph(column_index, part_pointers, part2Node, num_in_rows: int, num_out_rows: int, partSize: int,
                  dims=()) -> dict:
    """gnna_prepare_graph: counting pass, statistics, phase choice per width and scratch sizing up front (one stream
    synchronisation here, none in any later aggregation on this graph); the plan stays until the column_index storage
    is freed or `release_graph`.  CONTRACT: from here until the release, `column_index`, `part_pointers` and `part2Node`
    must not be rewritten in place -- the plan keeps a copy of the ids in the order the kernels read them (a sample
    checksum notices a rewritten buffer, not a few changed entries; `set_tuning(pack_ids=2)` turns the copy off).
    -> {dim: phases the library will use}."""
    dims = [int(d) for d in dims]
    arr = (ctypes.c_int * max(1, len(dims)))(*dims)
    out = (ctypes.c_int * max(1, len(dims)))()
    with torch.cuda.device(column_index.device):
        _check(load().gnna_prepare_graph(column_index.data_ptr(), part_pointers.data_ptr(), part2Node.data_ptr(),
                                         part2Node.numel(), int(num_in_rows), int(num_out_rows), int(partSize),
                                         arr, len(dims), out, _stream(column_index.device)))
    _forget_when_freed(column_index)
    return {d: int(out[i]) for i, d in enumerate(dims)}


def release_graph(column_index) -> None:
    """Drops the library's plans, hints and measured schedules for this graph (None: all graphs)."""
    ptr = None if column_index is None else column_index.data_ptr()
    _check(load().gnna_release_graph(ptr))
    if column_index is None:
        _registered.clear()
    else:
        _registered.pop(ptr, None)


def runtime_counters() -> dict:
    out = (ctypes.c_int64 * 16)()
    n = load().gnna_runtime_counters_ex(out, 16)
    names = ("plan_builds", "launch_syncs", "launch_frees", "launch_mallocs", "backoff_skips", "sweep_launches",
             "pack_builds", "packed_launches", "full_hashes")
    return {name: int(out[i]) for i, name in enumerate(names) if i < n}


def set_graph_phases(column_index, dim: int, column_phases: int) -> None:
    """Measured schedule for `dim`-wide aggregations on this graph (0 removes it); see include/gnna.h."""
    _check(load().gnna_set_graph_phases(column_index.data_ptr(), int(dim), int(column_phases)))
    if column_phases > 0:
        _forget_when_freed(column_index)


def xtg(X, G, out=None):
    """dW[K, N] = X^T G for X [M, K], G [M, N] (fp32, MFMA): the weight gradient of the dense update."""
    if not X.is_cuda:
        raise GnnaError("xtg needs device tensors: there is no CPU path in libgnna")
    assert X.dtype == torch.float32 and G.dtype == torch.float32 and X.dim() == 2 and G.dim() == 2
    assert X.shape[0] == G.shape[0]
    X, G = X.contiguous(), G.contiguous()
    if out is None:
        out = torch.empty(X.shape[1], G.shape[1], dtype=torch.float32, device=X.device)
    with torch.cuda.device(X.device):
        _check(load().gnna_xtg_f32(X.data_ptr(), G.data_ptr(), out.data_ptr(), X.shape[0], X.shape[1], G.shape[1],
                              _stream(X.device)))
    return out


def sddmm(dst_feat, src_feat, column_index, part_pointers, part2Node, partSize=32, out=None):
    """edge_out[e] = <dst_feat[row(e)], src_feat[column_index[e]]> over the neighbor-group partition
    (build-defined extension, see include/gnna.h).  Both feature matrices may be row-strided views (stride(1) == 1):
    their stride(0) is handed over as the leading dimension (gnna_sddmm_ld_f32)."""
    if not dst_feat.is_cuda:
        raise GnnaError("sddmm needs device tensors: there is no CPU path in libgnna")
    ap, n_out, dim, ld_dst = _rows_view(dst_feat, "dst_feat")
    bp, n_in, dim_b, ld_src = _rows_view(src_feat, "src_feat")
    assert dim == dim_b and dst_feat.device == src_feat.device
    if out is None:
        out = torch.zeros(column_index.numel(), dtype=torch.float32, device=dst_feat.device)
    with torch.cuda.device(dst_feat.device):
        if ld_dst == dim and ld_src == dim:
            _check(load().gnna_sddmm_f32(ap, bp, column_index.data_ptr(), part_pointers.data_ptr(), part2Node.data_ptr(),
                                         out.data_ptr(), n_out, n_in, dim, part2Node.numel(), int(partSize),
                                         _stream(dst_feat.device)))
        else:
            _check(load().gnna_sddmm_ld_f32(ap, ld_dst, bp, ld_src, column_index.data_ptr(), part_pointers.data_ptr(),
                                            part2Node.data_ptr(), out.data_ptr(), n_out, n_in, dim, part2Node.numel(),
                                            int(partSize), _stream(dst_feat.device)))
    return out
