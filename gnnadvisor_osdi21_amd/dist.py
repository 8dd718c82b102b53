"""Multi-GPU aggregation: destination-range sharding + RCCL all-gather of source features.

The reference is single-GPU (no distributed code at all); this module is
the MI355X design for BASELINE.json config 5.  One process per GPU (torch.distributed,
backend "nccl" == RCCL over xGMI):

* destination rows (CSR rows) are split into ``world`` contiguous blocks; rank r owns the
  CSR rows of block r, its block of X, and writes only its block of Y -- no collective on
  the outputs, no cross-GPU atomics.
* before an aggregation every rank needs the source rows it references: one
  ``all_gather_into_tensor`` of the [rows_per_rank, D] fp32 blocks (a single large
  collective; on the fully connected 8-GPU xGMI node every block moves one hop).
* blocks are padded to a common ``rows_per_rank`` so the gather lands directly in the
  layout the kernel reads; column ids are remapped once, at construction, into that
  padded global numbering -- no per-step unpadding copy.
* overlap: the shard's CSR is split once into the edges whose source is local (read
  straight from the rank's own block) and the edges whose source is remote.  Per step the
  all-gather is issued asynchronously, the local-source part is aggregated while the
  remote blocks arrive over xGMI, then the remote-source part is accumulated into the same
  output (``gnna_agg_rect_f32(..., accumulate=1)``).

* pipelined exchange (``pipeline_chunks = K``): the blocks are cut into K sub-blocks, the gather buffer is
  sub-block-major, K asynchronous all-gathers are in flight at once and the part of the remote CSR whose
  sources travel in piece k is aggregated as soon as piece k has arrived: the remote part is split into K small
  CSRs at construction, each an ordinary stateless aggregation whose column ids index the piece's own window of
  the receive buffer.  (The library's windowed entry ``gnna_agg_rect_windows_f32`` is the alternative for ONE CSR
  over K windows; since 0.4.0 it is stateless too -- per-window id counts instead of per-run cursors -- and
  refuses groups whose ids are not in increasing order, because a window call takes id POSITIONS.)
  ``pipeline_chunks = 0`` decides K from a time model of the exchange (``exposed_exchange_us``).
* halo exchange (``exchange="halo"``; ``"auto"`` picks it when it moves clearly fewer bytes): instead of
  whole blocks, every rank receives only the remote source rows its shard actually references.  The
  unique remote ids per owner are found once, the owners learn which of their rows each peer needs
  (one all-to-all of index lists at construction), and per step each rank gathers those rows
  (``index_select``) and the ranks swap them with ``all_to_all_single``; the remote part's column ids
  are remapped into the compact halo buffer.  The K-piece pipeline applies unchanged: piece k of every
  peer's list lands in window k of the buffer.  On an id-local (community-ordered) partition the halo is
  a small fraction of the all-gather volume; on a randomly labelled graph it is not, and ``"auto"``
  keeps the all-gather there.
* ``ShardedGCNConv`` / ``ShardedGINConv``: the layers on top (local dense update, sharded aggregation in
  forward and backward, all-reduce of the weight gradient); ``dist_main.py`` is the training driver.

``aggregate_fn`` is the local kernel (defaults to the HIP path through the C ABI); the
CPU/gloo tests inject a checker there, the product never does.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import torch
import torch.distributed as dist


def balanced_row_splits(row_pointers: torch.Tensor, world: int) -> list[int]:
    """Row boundaries [b_0=0, ..., b_world=N] balancing nnz (not node count) per block."""
    rp = row_pointers.to(torch.int64).cpu()
    n = rp.numel() - 1
    nnz = int(rp[-1])
    targets = torch.arange(1, world, dtype=torch.float64) * (nnz / world)
    cuts = torch.searchsorted(rp[1:].to(torch.float64), targets).tolist() if world > 1 else []
    bounds = [0]
    for c in cuts:
        bounds.append(int(min(max(c + 1, bounds[-1]), n)))
    bounds.append(n)
    return bounds


def shard_csr(row_pointers: torch.Tensor, column_index: torch.Tensor, lo: int, hi: int):
    """Rows [lo, hi) of a CSR -> (rebased int32 row_pointers, column slice with GLOBAL ids)."""
    rp = row_pointers.to(torch.int64)
    beg, end = int(rp[lo]), int(rp[hi])
    local_rp = (rp[lo:hi + 1] - beg).to(torch.int32)
    return local_rp, column_index[beg:end].contiguous()


def remap_columns_to_padded(column_index: torch.Tensor, bounds: Sequence[int], rows_per_rank: int,
                            chunks: int = 1):
    """Global node id -> position in the padded all-gather layout.

    chunks == 1: rank-major, owner_rank * rows_per_rank + (id - bounds[owner_rank]).
    chunks == K > 1 (pipelined exchange): every rank's padded block is cut into K sub-blocks of
    rows_per_rank / K rows and the buffer is sub-block-major -- position
    k * (world * rows_k) + owner_rank * rows_k + offset_in_sub_block -- so that the k-th
    all-gather (sub-block k of every rank) fills one contiguous slice of the buffer, which is
    exactly source window k of ``gnna_agg_rect_windows_f32``."""
    assert rows_per_rank % chunks == 0
    b = torch.as_tensor(list(bounds), dtype=torch.int64, device=column_index.device)
    world = len(bounds) - 1
    ci = column_index.to(torch.int64)
    owner = torch.searchsorted(b[1:], ci, right=True)
    off = ci - b[owner]
    if chunks == 1:
        out = owner * rows_per_rank + off
    else:
        rows_k = rows_per_rank // chunks
        k = torch.div(off, rows_k, rounding_mode="floor")
        out = k * (world * rows_k) + owner * rows_k + (off - k * rows_k)
    if int(out.max()) >= 2**31 if out.numel() else False:
        raise ValueError("padded node id exceeds int32")
    return out.to(torch.int32)


def sort_columns_within_rows(row_pointers: torch.Tensor, column_index: torch.Tensor) -> torch.Tensor:
    """Column ids of every CSR row in increasing order (the order the windowed / phased schedule
    consumes them in).  Needed after a remap that is not monotonic in the global id."""
    rp = row_pointers.to(torch.int64)
    n = rp.numel() - 1
    if column_index.numel() == 0:
        return column_index
    rows = torch.repeat_interleave(torch.arange(n, device=rp.device), rp[1:] - rp[:-1])
    span = int(column_index.max()) + 1
    key = rows * span + column_index.to(torch.int64)
    return (torch.sort(key).values - rows * span).to(column_index.dtype)


def _default_aggregate(mode, X_all, column_index, part_pointers, part2Node, num_out_rows, partSize,
                       degrees_out=None, degrees_in=None, epsilon=1.0, out=None, accumulate=False, windows=None):
    from . import _lib
    return _lib.agg_rect(mode, X_all, column_index, part_pointers, part2Node, num_out_rows, partSize,
                         degrees_out, degrees_in, epsilon, out, accumulate, windows=windows)


def exposed_exchange_us(world: int, nnz: int, rows_per_rank: int):
    """The time model behind the automatic piece count of the pipelined exchange (see ShardedAggregator): microseconds
    of an all-gather of every rank's `rows_per_rank` rows that neither the local-source aggregation hides nor the
    extra flushes of a piece-wise remote aggregation (+12 %) would eat up again; > 0 -> pipeline in pieces.
    Width-independent (256-byte rows on both sides).  -> (microseconds, remote edges of this shard)."""
    remote_edges = nnz * (world - 1) // max(1, world)
    local_edges = nnz - remote_edges
    received_rows = (world - 1) * rows_per_rank
    t_exchange = received_rows * 256.0 / (60e9 * max(1, min(world - 1, 7)))
    t_local = local_edges / 60e9
    t_remote = remote_edges / 45e9
    return int(1e6 * (t_exchange - t_local - 0.12 * t_remote)), remote_edges


def _default_hints(column_index: torch.Tensor, avg_degree: float, nonlocal_ids: bool) -> None:
    """Tell libgnna what kind of CSR this column_index array belongs to (enables its column-phased
    schedule for high-degree parts whose sources are scattered).  Registered once per part."""
    from . import _lib
    if column_index.is_cuda and column_index.numel():
        _lib.set_graph_hints(column_index, max(1, int(avg_degree)), nonlocal_ids)


def split_local_remote(local_row_pointers: torch.Tensor, column_index: torch.Tensor, lo: int, hi: int):
    """Split a shard's CSR (global column ids) into the part whose sources lie in [lo, hi)
    (returned with ids rebased to the local block) and the rest (ids unchanged).
    -> (rp_local, ci_local, rp_remote, ci_remote), row pointers int32 on the input's device."""
    rp = local_row_pointers.to(torch.int64)
    n = rp.numel() - 1
    deg = rp[1:] - rp[:-1]
    rows = torch.repeat_interleave(torch.arange(n, device=rp.device), deg)
    ci = column_index.to(torch.int64)
    is_local = (ci >= lo) & (ci < hi)

    def csr_of(mask):
        r = torch.zeros(n + 1, dtype=torch.int64, device=rp.device)
        r[1:] = torch.cumsum(torch.bincount(rows[mask], minlength=n), 0)
        return r.to(torch.int32)

    return (csr_of(is_local), (ci[is_local] - lo).to(torch.int32),
            csr_of(~is_local), ci[~is_local].to(torch.int32))


class ShardedAggregator:
    """Aggregation over one destination-range shard of a graph.

    Parameters
    ----------
    local_row_pointers : int32 [n_local + 1]   CSR rows owned by this rank (rebased to 0)
    column_index       : int32 [nnz_local]     GLOBAL source ids of those rows
    bounds             : world + 1 row boundaries (``balanced_row_splits``)
    partSize           : neighbor-group size for the local partition
    """

    def __init__(self, local_row_pointers: torch.Tensor, column_index: torch.Tensor,
                 bounds: Sequence[int], partSize: int = 32, *, group=None, device=None,
                 aggregate_fn: Optional[Callable] = None, build_part_fn: Optional[Callable] = None,
                 overlap: bool = True, force_overlap: bool = False,
                 hint_fn: Optional[Callable] = None, scattered_sources: bool = True,
                 pipeline_chunks: int = 0, exchange: str = "allgather", emulate: Optional[tuple] = None,
                 force_collectives: bool = False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # force_collectives (debug / bring-up): with ONE rank every collective of the N-rank step is still issued
        # through the process group -- all_gather_into_tensor (whole and K pieces into adjacent views),
        # all_to_all_single with split lists, the all-reduces of the set-up decisions and of dW -- so that the RCCL
        # path can be executed, ordered against the library's launches and checked on a single GPU.  The rank's own
        # block then plays both roles: sources in its first half are "local", sources in its second half are
        # "remote" and travel through the collective (to the rank itself) before the remote part reads them.
        self.force_collectives = bool(force_collectives) and dist.is_initialized() and self.world == 1 and emulate is None
        # emulate = (rank, world): this process plays ONE rank of a `world`-rank job without a process group -- the
        # shard, its local / remote split, the halo lists and the piece CSRs are built exactly as that rank would
        # build them; only the collectives are missing (the receive buffers are filled by `emulated_receive`).
        # Used to run BASELINE config 5's true per-rank shape on one GPU (bench.py, tests).
        self.emulated = emulate is not None
        if self.emulated:
            assert not dist.is_initialized() or dist.get_world_size(group) == 1
            self.rank, self.world = int(emulate[0]), int(emulate[1])
        assert len(bounds) == self.world + 1
        self.bounds = [int(b) for b in bounds]
        self.n_local = self.bounds[self.rank + 1] - self.bounds[self.rank]
        assert local_row_pointers.numel() == self.n_local + 1
        self.rows_per_rank = max(1, max(self.bounds[i + 1] - self.bounds[i] for i in range(self.world)))
        self.overlap = bool(overlap) and (self.world > 1 or force_overlap or self.force_collectives)
        # collectives: does a step talk to the process group at all
        self.collectives = (self.world > 1 and emulate is None) or self.force_collectives
        # pipelined exchange: K all-gathers of 1/K of every block, each aggregated on arrival
        # (0 = automatic: 4 pieces once the remote part is big enough to keep every piece busy)
        assert exchange in ("allgather", "halo", "auto")
        self.device = torch.device(device) if device is not None else column_index.device
        if exchange != "allgather" and (self.world > 1 or self.force_collectives):
            self.overlap = True                                  # the halo buffer holds remote rows only
        K = int(pipeline_chunks)
        if K <= 0:
            # Pipelining the exchange pays when the wire time the local-source part cannot hide exceeds what the pieces
            # cost: K piece calls flush every destination row K times (one-GPU emulation of Reddit-sized shards,
            # tools/probe_dist_emul.py: remote part in one call / in 4 pieces 1.01 / 1.12 ms at 2 ranks, 1.62 / 1.91
            # at 4, 2.28 / 2.36 at 8 -- taken as +12 %).  A time model, in units that do not depend on the feature
            # width (both sides scale with it; 256-byte rows): the exchange moves `received_rows` rows over
            # min(world - 1, 7) xGMI links at ~60 GB/s each (76.8 GB/s per direction, RCCL reaches ~80 %); the
            # local part gathers ~60 G edges/s, the remote part, whose source matrix is world times larger, ~45.
            #   2 ranks: 0.99 ms of exchange behind 0.95 ms of local work -> one piece;
            #   4 ranks: 0.99 vs 0.48 ms, the remaining 0.5 ms exceed the pieces' 0.23 ms -> four pieces;
            #   8 ranks: 0.99 vs 0.24 ms -> four pieces (2.7 instead of ~3.3 ms per step in the model);
            #   papers100M-like shards (14.5 edges per row): the exchange dwarfs the aggregation -> four pieces.
            # The constants are estimates until a multi-GPU run measures the exchange (`exchange_only_ms` in the
            # bench line).  Every rank must arrive at the same K (it fixes the layout and the number of collectives
            # per step), so the decision is taken collectively on the most exchange-bound and the largest shard.
            mine_us, remote_edges = exposed_exchange_us(self.world, int(column_index.numel()), self.rows_per_rank)
            exposed_us = self._agree_max(mine_us)
            largest = self._agree_max(remote_edges)
            K = 4 if (self.world > 1 and largest >= (16 << 20) and exposed_us > 0) else 1
            if self.force_collectives:
                K = 1      # (one rank: nothing to hide; ask for pieces explicitly with pipeline_chunks)
        self.chunks = max(1, min(K, 16, self.rows_per_rank)) if self.overlap else 1
        assert self._agree_max(self.chunks) == self.chunks == -self._agree_max(-self.chunks), \
            "ranks disagree on the number of exchange pieces"
        self.chunk_rows = (self.rows_per_rank + self.chunks - 1) // self.chunks
        self.rows_per_rank = self.chunk_rows * self.chunks          # padded so the pieces are equal
        self.partSize = int(partSize)
        # hints are only meaningful for the real kernel; an injected aggregate_fn gets none
        self.hint_fn = hint_fn if hint_fn is not None else (_default_hints if aggregate_fn is None else None)
        self.scattered_sources = bool(scattered_sources)
        self.aggregate_fn = aggregate_fn or _default_aggregate
        if build_part_fn is None:
            from . import _lib
            build_part_fn = _lib.build_part
        pp, p2n = build_part_fn(self.partSize, local_row_pointers.cpu().contiguous())
        self.row_pointers = local_row_pointers.to(self.device)
        self.column_index = remap_columns_to_padded(column_index.to(self.device), self.bounds,
                                                    self.rows_per_rank, self.chunks)
        if self.chunks > 1:
            self.column_index = sort_columns_within_rows(self.row_pointers, self.column_index)
        self.part_pointers = pp.to(self.device)
        self.part2Node = p2n.to(self.device)
        self.halo_rows = 0
        self.remote_rows = self.world * self.rows_per_rank
        # local-source / remote-source split for the overlapped schedule
        if self.overlap:
            lo, hi = self.bounds[self.rank], self.bounds[self.rank + 1]
            if self.force_collectives:
                hi = lo + (hi - lo) // 2                         # the second half of the own block is "remote"
            self.local_rows = hi - lo                            # source rows the local part reads (of X_local)
            rp_l, ci_l, rp_r, ci_r = split_local_remote(local_row_pointers.to(self.device),
                                                        column_index.to(self.device), lo, hi)
            pp_l, p2n_l = build_part_fn(self.partSize, rp_l.cpu().contiguous())
            pp_r, p2n_r = build_part_fn(self.partSize, rp_r.cpu().contiguous())
            self.avg_degree_local = ci_l.numel() / max(1, self.n_local)
            self.avg_degree_remote = ci_r.numel() / max(1, self.n_local)
            self.local_part = (ci_l.contiguous(), pp_l.to(self.device), p2n_l.to(self.device))
            self.exchange = self._plan_exchange(exchange, ci_r)
            if self.exchange == "halo":
                ci_r = self._halo_remap(ci_r)
            else:
                ci_r = remap_columns_to_padded(ci_r, self.bounds, self.rows_per_rank, self.chunks)
            if self.chunks > 1:
                ci_r = sort_columns_within_rows(rp_r, ci_r)
            self.remote_part = (ci_r.contiguous(), pp_r.to(self.device), p2n_r.to(self.device))
            # pipelined exchange: the remote part is split once more, by the piece of the exchange its source arrives
            # with; piece k is then an ordinary aggregation (accumulate) over its own small CSR -- it reads nothing but
            # rows of piece k by construction, runs on the streaming kernel and keeps no state between the K calls.
            # Its column ids are relative to the piece's window of the receive buffer and the call gets that window as
            # its source matrix: the library then slices 1/K of the buffer, not K times the range the ids cover
            # (which made 1 - 1/K of a piece's phases empty passes over its descriptors).
            self.remote_pieces = []
            self.piece_rows = 0
            if self.chunks > 1:
                win = self.remote_rows // self.chunks
                self.piece_rows = win
                rows_r = torch.repeat_interleave(torch.arange(self.n_local, device=self.device),
                                                 (rp_r[1:] - rp_r[:-1]).to(torch.int64).to(self.device))
                piece_of = torch.div(ci_r.to(torch.int64), win, rounding_mode="floor")
                for k in range(self.chunks):
                    m = piece_of == k
                    rp_k = torch.zeros(self.n_local + 1, dtype=torch.int64, device=self.device)
                    rp_k[1:] = torch.cumsum(torch.bincount(rows_r[m], minlength=self.n_local), 0)
                    pp_k, p2n_k = build_part_fn(self.partSize, rp_k.to(torch.int32).cpu().contiguous())
                    self.remote_pieces.append(((ci_r[m] - k * win).to(torch.int32).contiguous(), pp_k.to(self.device),
                                               p2n_k.to(self.device)))
        else:
            self.exchange = "allgather"
        self.avg_degree_all = column_index.numel() / max(1, self.n_local)
        if self.hint_fn:
            self.hint_fn(self.column_index, self.avg_degree_all, self.scattered_sources)
            if self.overlap:
                self.hint_fn(self.local_part[0], self.avg_degree_local, self.scattered_sources)
                self.hint_fn(self.remote_part[0], self.avg_degree_remote, self.scattered_sources)
                for ci_k, _, _ in self.remote_pieces:
                    self.hint_fn(ci_k, self.avg_degree_remote / self.chunks, self.scattered_sources)
        self._gather_buf: Optional[torch.Tensor] = None
        self._pad_buf: Optional[torch.Tensor] = None
        self._deg_all: Optional[torch.Tensor] = None
        self._deg_src: Optional[torch.Tensor] = None
        self._halo_buf: Optional[torch.Tensor] = None

    @property
    def nnz_local(self) -> int:
        return int(self.column_index.numel())

    def release(self) -> None:
        """Gives back what this aggregator holds: its buffers and index arrays, and -- for arrays on the GPU that the library
        has seen -- the library's plans and hints keyed by them (an address the caching allocator hands to the next graph
        must not find the old graph's plan).  The object is unusable afterwards."""
        arrays = [getattr(self, "column_index", None)]
        for part in (getattr(self, "local_part", None), getattr(self, "remote_part", None), *getattr(self, "remote_pieces", [])):
            if part is not None:
                arrays.append(part[0])
        if self.aggregate_fn is _default_aggregate:
            from . import _lib
            for a in arrays:
                if a is not None and a.is_cuda and a.numel():
                    _lib.release_graph(a)
        self._gather_buf = self._pad_buf = self._deg_all = self._deg_src = self._halo_buf = None
        self.column_index = self.part_pointers = self.part2Node = self.row_pointers = None
        self.local_part = self.remote_part = None
        self.remote_pieces = []
        self._in_flight = None

    # ---- collective set-up decisions ------------------------------------------------------------
    def _agree_max(self, value: int) -> int:
        """max of an integer over the ranks of the group (identity without a process group)."""
        if self.emulated or not (dist.is_initialized() and (self.world > 1 or self.force_collectives)):
            return int(value)
        on_gpu = dist.get_backend(self.group) == "nccl"
        t = torch.tensor([int(value)], dtype=torch.int64, device=self.device if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return int(t.item())

    def _comm_device(self):
        return self.device if (dist.is_initialized() and dist.get_backend(self.group) == "nccl") else torch.device("cpu")

    def _plan_exchange(self, exchange: str, ci_remote_global: torch.Tensor) -> str:
        """Decides between the block all-gather and the halo exchange (collectively: every rank must take the
        same path) and, for the halo exchange, builds its index lists.  ``ci_remote_global``: global source
        ids of the remote-source edges of this shard."""
        self.halo_rows = 0
        self.remote_rows = self.world * self.rows_per_rank       # rows of the buffer the remote part reads
        if exchange == "allgather" or not (self.world > 1 or self.force_collectives):
            return "allgather"
        uniq = torch.unique(ci_remote_global.to(torch.int64))    # sorted => grouped by owner
        worst = self._agree_max(int(uniq.numel()))
        # rows a peer set could ask for at most (one forced rank: the "remote" half of its own block)
        candidates = (self.world - 1) * self.rows_per_rank if self.world > 1 else self.n_local - self.n_local // 2
        if exchange == "auto" and worst > 0.7 * candidates:
            return "allgather"                                   # nearly everything is referenced anyway
        K, world = self.chunks, self.world
        b = torch.as_tensor(self.bounds, dtype=torch.int64, device=uniq.device)
        owner = torch.searchsorted(b[1:], uniq, right=True)
        need = torch.bincount(owner, minlength=world)            # rows wanted from each peer
        need_l = need.tolist()
        if self.emulated:
            # no peers to ask.  By symmetry of the generator the rows the peers want from this rank look like the rows this
            # rank wants from them (offsets inside the owner's block): they stand in for the send lists, so that the
            # send-side gather (index_select per piece) can be timed too; the aggregation itself does not use them
            send_l, asked = list(need_l), (uniq - b[owner]).clamp_(0, max(0, self.n_local - 1)).to(self.device)
        else:
            cdev = self._comm_device()
            send_counts = torch.empty(world, dtype=torch.int64, device=cdev)
            dist.all_to_all_single(send_counts, need.to(cdev), group=self.group)
            send_l = send_counts.tolist()
            wanted = (uniq - b[owner]).to(cdev)                      # offsets inside the owner's block
            asked = torch.empty(int(sum(send_l)), dtype=torch.int64, device=cdev)
            dist.all_to_all_single(asked, wanted, output_split_sizes=send_l, input_split_sizes=need_l, group=self.group)
            asked = asked.to(self.device)
            assert not asked.numel() or (int(asked.min()) >= 0 and int(asked.max()) < self.n_local)
        self.halo_rows_per_peer = need_l

        # piece k of a list of n rows = rows [k*c, (k+1)*c) with c = ceil(n / K); both sides derive it from n
        def pieces(n):
            c = (n + K - 1) // K if n else 0
            return [max(0, min(n, (k + 1) * c) - min(n, k * c)) for k in range(K)]
        need_pc = [pieces(n) for n in need_l]                    # [peer][k]
        send_pc = [pieces(n) for n in send_l]
        self._recv_splits = [[need_pc[p][k] for p in range(world)] for k in range(K)]
        self._send_splits = [[send_pc[p][k] for p in range(world)] for k in range(K)]
        self.window_rows = max(1, max(sum(r) for r in self._recv_splits))
        self.remote_rows = K * self.window_rows
        # send side: per piece, the rows of this rank's block to ship, concatenated by receiver
        starts = [0]
        for n in send_l:
            starts.append(starts[-1] + n)
        self._send_index = []
        for k in range(K):
            parts = []
            for p in range(world):
                c = (send_l[p] + K - 1) // K if send_l[p] else 0
                lo = starts[p] + min(send_l[p], k * c)
                parts.append(asked[lo: lo + send_pc[p][k]])
            self._send_index.append(torch.cat(parts) if parts else asked[:0])
        # receive side: position of every needed id in the window-major halo buffer
        pos = torch.empty_like(uniq)
        at = 0
        for p in range(world):
            n = need_l[p]
            if not n:
                continue
            c = (n + K - 1) // K
            j = torch.arange(n, device=uniq.device)
            k = torch.div(j, c, rounding_mode="floor")
            off = torch.zeros(K, dtype=torch.int64, device=uniq.device)
            for kk in range(K):
                off[kk] = kk * self.window_rows + sum(self._recv_splits[kk][:p])
            pos[at: at + n] = off[k] + (j - k * c)
            at += n
        self._halo_ids, self._halo_pos = uniq, pos
        self.halo_rows = int(uniq.numel())
        return "halo"

    def _halo_remap(self, ci_remote_global: torch.Tensor) -> torch.Tensor:
        idx = torch.searchsorted(self._halo_ids, ci_remote_global.to(torch.int64))
        return self._halo_pos[idx].to(torch.int32)

    def exchange_halo(self, X_local: torch.Tensor, buf: Optional[torch.Tensor] = None):
        """Ships the rows of this rank's block that the peers reference and receives this shard's halo rows
        into the window-major buffer: K asynchronous all_to_all_single calls.  -> (buffer, [work handles])."""
        D = X_local.shape[1] if X_local.dim() == 2 else 1
        shape = (self.remote_rows, D) if X_local.dim() == 2 else (self.remote_rows,)
        if buf is None:
            if self._halo_buf is None or self._halo_buf.shape != shape or self._halo_buf.device != X_local.device:
                self._halo_buf = torch.zeros(shape, dtype=X_local.dtype, device=X_local.device)
            buf = self._halo_buf
        works, keep = [], []
        for k in range(self.chunks):
            send = X_local.index_select(0, self._send_index[k])
            n_in = sum(self._recv_splits[k])
            recv = buf[k * self.window_rows: k * self.window_rows + n_in]
            keep.append(send)
            if send.is_cuda and self._comm_device().type == "cpu":
                # debug / test path (gloo moves host memory): stage the piece through the host, synchronously
                got = torch.empty(recv.shape, dtype=recv.dtype)
                dist.all_to_all_single(got, send.cpu(), output_split_sizes=self._recv_splits[k],
                                       input_split_sizes=self._send_splits[k], group=self.group)
                recv.copy_(got)
                works.append(None)
                continue
            works.append(dist.all_to_all_single(recv, send, output_split_sizes=self._recv_splits[k],
                                                input_split_sizes=self._send_splits[k], group=self.group,
                                                async_op=True))
        self._in_flight = keep                                   # the send buffers live until the next exchange
        return buf, works

    def emulated_receive(self, X_global: torch.Tensor) -> torch.Tensor:
        """Emulation only: puts into this rank's receive buffer what the exchange would have delivered, from the
        features of ALL ranks in global node order (row bounds[p] + i = row i of rank p).  -> the buffer the
        remote part reads (the all-gather buffer in its padded / sub-block-major layout, or the compact halo
        buffer)."""
        assert self.emulated and X_global.shape[0] == self.bounds[-1]
        D = X_global.shape[1]
        if self.exchange == "halo":
            buf = torch.zeros(self.remote_rows, D, dtype=X_global.dtype, device=X_global.device)
            step = 1 << 22
            for a in range(0, int(self._halo_ids.numel()), step):        # in slabs: bounded temporaries
                buf.index_copy_(0, self._halo_pos[a:a + step], X_global.index_select(0, self._halo_ids[a:a + step]))
            self._halo_buf = buf
            return buf
        even = all(self.bounds[i + 1] - self.bounds[i] == self.rows_per_rank for i in range(self.world))
        if self.chunks == 1 and even:
            self._gather_buf = X_global                                  # the padded rank-major layout IS global order
            return X_global
        buf = torch.zeros(self.world * self.rows_per_rank, D, dtype=X_global.dtype, device=X_global.device)
        rk = self.chunk_rows
        for pr in range(self.world):
            lo, hi = self.bounds[pr], self.bounds[pr + 1]
            for k in range(self.chunks):
                a, b2 = lo + k * rk, min(hi, lo + (k + 1) * rk)
                if b2 > a:
                    at = k * self.world * rk + pr * rk if self.chunks > 1 else pr * self.rows_per_rank
                    buf[at: at + (b2 - a)].copy_(X_global[a:b2])
        self._gather_buf = buf
        return buf

    def send_side_gather(self, X_local: torch.Tensor):
        """The send half of one halo exchange without the collective: per piece, the rows of this rank's block that the
        peers reference, gathered into a contiguous send buffer (what `exchange_halo` does before `all_to_all_single`).
        -> list of the K send buffers."""
        assert self.exchange == "halo"
        return [X_local.index_select(0, self._send_index[k]) for k in range(self.chunks)]

    def bytes_received_per_step(self, dim: int) -> int:
        """Feature bytes this rank receives from its peers per aggregation (forced one-rank collectives: the bytes the
        rank sends to itself)."""
        if not (self.world > 1 or self.force_collectives):
            return 0
        if self.exchange == "halo":
            return self.halo_rows * dim * 4
        return max(1, self.world - 1) * self.rows_per_rank * dim * 4

    def allgather_bytes_per_step(self, dim: int) -> int:
        if self.force_collectives:
            return self.rows_per_rank * dim * 4
        return (self.world - 1) * self.rows_per_rank * dim * 4 if self.world > 1 else 0

    def describe_exchange(self) -> str:
        if not (self.world > 1 or self.force_collectives):
            return "no exchange (one rank)"
        how = ("halo rows only (all_to_all_single of %d of %d remote rows)" % (self.halo_rows, (self.world - 1) * self.rows_per_rank)
               if self.exchange == "halo" else "all-gather of the feature blocks")
        return f"{how} in {self.chunks} piece(s), overlapped with the local-source aggregation"

    def gather_features(self, X_local: torch.Tensor, async_op: bool = False):
        """all-gather the per-rank feature blocks into the padded [world * rows_per_rank, D] layout.
        With async_op the collective's work handle is returned as well: (buffer, work)."""
        assert X_local.shape[0] == self.n_local
        D = X_local.shape[1]
        if not self.collectives:
            return (X_local, None) if async_op else X_local
        shape = (self.world * self.rows_per_rank, D)
        if self._gather_buf is None or self._gather_buf.shape != shape or self._gather_buf.device != X_local.device:
            self._gather_buf = torch.empty(shape, dtype=X_local.dtype, device=X_local.device)
        src = X_local
        if self.n_local != self.rows_per_rank:
            if self._pad_buf is None or self._pad_buf.shape != (self.rows_per_rank, D):
                self._pad_buf = torch.zeros(self.rows_per_rank, D, dtype=X_local.dtype, device=X_local.device)
            self._pad_buf[: self.n_local].copy_(X_local)
            src = self._pad_buf
        work = dist.all_gather_into_tensor(self._gather_buf, src.contiguous(), group=self.group,
                                           async_op=async_op)
        return (self._gather_buf, work) if async_op else self._gather_buf

    def gather_feature_chunks(self, X_local: torch.Tensor):
        """Pipelined exchange: K asynchronous all-gathers, the k-th moving sub-block k of every rank's
        (padded) block into slice k of the sub-block-major buffer.  -> (buffer, [work handles])."""
        assert X_local.shape[0] == self.n_local
        D, K, rk = X_local.shape[1], self.chunks, self.chunk_rows
        src = X_local
        if self.n_local != self.rows_per_rank:
            if self._pad_buf is None or self._pad_buf.shape != (self.rows_per_rank, D):
                self._pad_buf = torch.zeros(self.rows_per_rank, D, dtype=X_local.dtype, device=X_local.device)
            self._pad_buf[: self.n_local].copy_(X_local)
            src = self._pad_buf
        if not self.collectives:
            return src, [None] * K
        shape = (self.world * self.rows_per_rank, D)
        if self._gather_buf is None or self._gather_buf.shape != shape or self._gather_buf.device != X_local.device:
            self._gather_buf = torch.empty(shape, dtype=X_local.dtype, device=X_local.device)
        works = []
        for k in range(K):
            dst = self._gather_buf[k * self.world * rk: (k + 1) * self.world * rk]
            works.append(dist.all_gather_into_tensor(dst, src[k * rk: (k + 1) * rk], group=self.group,
                                                     async_op=True))
        return self._gather_buf, works

    def prepare_degrees(self, degrees_local: torch.Tensor) -> torch.Tensor:
        """all-gather the per-node degree norms once (graph constant) into the padded layout."""
        assert degrees_local.numel() == self.n_local
        if self.exchange == "halo":
            buf = torch.ones(self.remote_rows, dtype=degrees_local.dtype, device=degrees_local.device)
            _, works = self.exchange_halo(degrees_local.contiguous(), buf)
            for w in works:
                if w is not None:
                    w.wait()
            self._deg_all = buf
        elif not self.collectives and self.chunks == 1:
            self._deg_all = degrees_local
        elif not self.collectives:
            pad = torch.ones(self.rows_per_rank, dtype=degrees_local.dtype, device=degrees_local.device)
            pad[: self.n_local] = degrees_local
            self._deg_all = pad
        else:
            pad = torch.ones(self.rows_per_rank, dtype=degrees_local.dtype, device=degrees_local.device)
            pad[: self.n_local] = degrees_local
            buf = torch.empty(self.world * self.rows_per_rank, dtype=degrees_local.dtype,
                              device=degrees_local.device)
            dist.all_gather_into_tensor(buf, pad, group=self.group)
            if self.chunks > 1:                                   # rank-major -> sub-block-major
                buf = buf.view(self.world, self.chunks, self.chunk_rows).permute(1, 0, 2).reshape(-1).contiguous()
            self._deg_all = buf
        self._deg_src = degrees_local
        return self._deg_all

    def aggregate(self, X_local: torch.Tensor, mode: int = 0, *, degrees_local=None, epsilon: float = 1.0,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Y_local = A[rows of this rank, :] @ X   (mode 0 sag / 1 gcn / 2 gin)."""
        deg_in = None
        if mode == 1:
            assert degrees_local is not None
            if self._deg_all is None or self._deg_src is not degrees_local:
                self.prepare_degrees(degrees_local)
            deg_in = self._deg_all
        if not self.overlap:
            X_all = self.gather_features(X_local)
            return self.aggregate_fn(mode, X_all, self.column_index, self.part_pointers, self.part2Node,
                                     self.n_local, self.partSize, degrees_local, deg_in, epsilon, out)
        if self.chunks > 1 or self.exchange == "halo":
            return self._aggregate_pipelined(X_local, mode, degrees_local, deg_in, epsilon, out)
        # overlapped: remote blocks travel while the local-source edges are aggregated
        X_all, work = self.gather_features(X_local, async_op=True)
        ci_l, pp_l, p2n_l = self.local_part
        out = self.aggregate_fn(mode, X_local, ci_l, pp_l, p2n_l, self.n_local, self.partSize,
                                degrees_local, degrees_local, epsilon, out)
        if work is not None:
            work.wait()
        ci_r, pp_r, p2n_r = self.remote_part
        return self.aggregate_fn(mode, X_all, ci_r, pp_r, p2n_r, self.n_local, self.partSize,
                                 degrees_local, deg_in, epsilon, out, accumulate=True)

    def _aggregate_pipelined(self, X_local, mode, degrees_local, deg_in, epsilon, out):
        """K-piece exchange: all pieces are put in flight at once; the local-source edges are
        aggregated meanwhile, then source window k of the remote part as soon as piece k is there."""
        if self.exchange == "halo":
            X_all, works = self.exchange_halo(X_local.contiguous())
        else:
            X_all, works = self.gather_feature_chunks(X_local)
        ci_l, pp_l, p2n_l = self.local_part
        out = self.aggregate_fn(mode, X_local, ci_l, pp_l, p2n_l, self.n_local, self.partSize,
                                degrees_local, degrees_local, epsilon, out)
        parts = self.remote_pieces if self.chunks > 1 else [self.remote_part]
        for k, ((ci_k, pp_k, p2n_k), work) in enumerate(zip(parts, works)):
            if work is not None:
                work.wait()
            if ci_k.numel():
                X_k, deg_k = self._piece_window(X_all, deg_in, k)
                out = self.aggregate_fn(mode, X_k, ci_k, pp_k, p2n_k, self.n_local, self.partSize,
                                        degrees_local, deg_k, epsilon, out, accumulate=True)
        return out

    def _piece_window(self, X_all, deg_in, k):
        """Source rows (and their degree norms) of exchange piece k: the window of the receive buffer its ids index."""
        if self.chunks <= 1:
            return X_all, deg_in
        lo, hi = k * self.piece_rows, (k + 1) * self.piece_rows
        return X_all[lo:hi], (deg_in[lo:hi] if deg_in is not None else None)

    # ---- the two halves of a step on their own (bench.py reports them beside the overlapped step) ----------------
    def exchange_only(self, X_local: torch.Tensor) -> None:
        """The source-feature exchange of one aggregation, waited for, without any aggregation."""
        if not self.collectives:
            return
        if self.exchange == "halo":
            _, works = self.exchange_halo(X_local.contiguous())
        elif self.chunks > 1:
            _, works = self.gather_feature_chunks(X_local)
        else:
            _, work = self.gather_features(X_local, async_op=True)
            works = [work]
        for w in works:
            if w is not None:
                w.wait()

    def aggregate_only(self, X_local: torch.Tensor, out: Optional[torch.Tensor] = None, mode: int = 0,
                       degrees_local: Optional[torch.Tensor] = None, epsilon: float = 1.0) -> torch.Tensor:
        """The kernels of one aggregation on the buffers as the last exchange (or `emulated_receive`) left them: no
        collective.  mode 1 (gcn) reads the degree norms of the sources from the buffer `prepare_degrees` /
        `emulated_receive_degrees` filled."""
        deg_in = self._deg_all if mode == 1 else None
        if mode == 1:
            assert degrees_local is not None and deg_in is not None
        if not self.overlap:
            X_all = X_local if not (self.collectives or self.emulated) else self._gather_buf
            return self.aggregate_fn(mode, X_all, self.column_index, self.part_pointers, self.part2Node, self.n_local,
                                     self.partSize, degrees_local, deg_in, epsilon, out)
        X_all = self._halo_buf if self.exchange == "halo" else (self._gather_buf if (self.collectives or self.emulated) else
                                                                (self._pad_buf if self._pad_buf is not None else X_local))
        ci_l, pp_l, p2n_l = self.local_part
        out = self.aggregate_fn(mode, X_local, ci_l, pp_l, p2n_l, self.n_local, self.partSize, degrees_local,
                                degrees_local, epsilon, out)
        if X_all is None:
            return out
        for k, (ci_k, pp_k, p2n_k) in enumerate(self.remote_pieces if self.chunks > 1 else [self.remote_part]):
            if ci_k.numel():
                X_k, deg_k = self._piece_window(X_all, deg_in, k)
                out = self.aggregate_fn(mode, X_k, ci_k, pp_k, p2n_k, self.n_local,
                                        self.partSize, degrees_local, deg_k, epsilon, out, accumulate=True)
        return out

    def emulated_receive_degrees(self, deg_global: torch.Tensor) -> torch.Tensor:
        """Emulation only: the degree norms of all nodes, in global order, laid out like the feature buffer."""
        keep = (self._gather_buf, self._halo_buf)
        buf = self.emulated_receive(deg_global.reshape(-1, 1).contiguous()).reshape(-1)
        self._gather_buf, self._halo_buf = keep
        self._deg_all = buf
        return buf

    def _parts(self):
        """(column ids, part pointers, part2Node, source rows) of every CSR this shard aggregates with."""
        n_all = self.remote_rows if self.overlap else self.world * self.rows_per_rank
        if not self.overlap:
            return [(self.column_index, self.part_pointers, self.part2Node, n_all if self.world > 1 else self.n_local)]
        parts = [self.local_part + (self.n_local,)]                  # (reads X_local: n_local rows)
        if self.chunks == 1:
            parts.append(self.remote_part + (n_all,))
        else:
            parts += [pc + (self.piece_rows,) for pc in self.remote_pieces]
        return [pt for pt in parts if pt[0].numel()]

    def prepare(self, dims) -> None:
        """Graph lifecycle for the shard's CSRs (``gnna_prepare_graph``): plans pinned, phase counts chosen and the
        column ids packed for the widths in ``dims`` up front -- the CSRs of a shard never change.  Only with the real
        kernel on a GPU."""
        if self.aggregate_fn is not _default_aggregate or self.device.type != "cuda":
            return
        from . import _lib
        for ci, pp, p2n, n_in in self._parts():
            _lib.prepare_graph(ci, pp, p2n, int(n_in), self.n_local, self.partSize, [int(d) for d in dims])

    def calibrate(self, dims, reps: int = 3) -> dict:
        """Measured phase counts for the parts of this shard (no collective involved: every rank tunes its
        own kernels on random features).  Every part -- local, remote, or the K pieces of the remote part -- goes
        through ``decider.calibrate_phases``.  Only with the real kernel (no injected aggregate_fn)."""
        if self.aggregate_fn is not _default_aggregate or self.device.type != "cuda":
            return {}
        from . import _lib
        from .decider import calibrate_phases
        res = {}
        self.prepare(dims)
        try:
            return self._calibrate(dims, res, calibrate_phases)
        finally:
            self.prepare(dims)       # the measured schedule's phase counts get their packed copies now

    def _calibrate(self, dims, res, calibrate_phases) -> dict:
        n_all = self.remote_rows if self.overlap else self.world * self.rows_per_rank
        if not self.overlap:
            res["whole"] = calibrate_phases(self.column_index, self.part_pointers, self.part2Node, self.n_local,
                                            self.partSize, dims, num_in_rows=n_all if self.world > 1 else self.n_local)
            return res
        ci_l, pp_l, p2n_l = self.local_part
        if ci_l.numel():
            res["local"] = calibrate_phases(ci_l, pp_l, p2n_l, self.n_local, self.partSize, dims, num_in_rows=self.n_local)
        ci_r, pp_r, p2n_r = self.remote_part
        if not ci_r.numel():
            return res
        if self.chunks == 1:
            res["remote"] = calibrate_phases(ci_r, pp_r, p2n_r, self.n_local, self.partSize, dims, num_in_rows=n_all)
            return res
        res["remote"] = [calibrate_phases(ci_k, pp_k, p2n_k, self.n_local, self.partSize, dims, num_in_rows=self.piece_rows)
                         if ci_k.numel() else {} for ci_k, pp_k, p2n_k in self.remote_pieces]
        return res

    def sag(self, X_local: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        return self.aggregate(X_local, 0, out=out)

    def gcn(self, X_local: torch.Tensor, degrees_local: torch.Tensor, out=None) -> torch.Tensor:
        return self.aggregate(X_local, 1, degrees_local=degrees_local, out=out)

    def gin(self, X_local: torch.Tensor, epsilon: float, out=None) -> torch.Tensor:
        return self.aggregate(X_local, 2, epsilon=epsilon, out=out)


# ---- sharded operator layer (SURVEY.md 8e "backward") ---------------------------------------
# Counterparts of ops.GNNAFunction / ops.GNNAFunction_GIN for one destination-range shard.  The
# dense update stays local (rows of X W only need the rank's own rows); the aggregation is the
# sharded one above; A is symmetric (as the reference assumes, gnn_conv.py:57-78), so the
# gradient aggregation uses the same shard; dW = X_local^T G_local is a partial sum over the
# rank's rows and is all-reduced (a [Fin, Fout] fp32 payload: latency-, not bandwidth-bound).

_trace_sink: Optional[list] = None


def timed_aggregator(local_row_pointers: torch.Tensor, column_index: torch.Tensor, bounds: Sequence[int], partSize: int = 32, *,
                     dim: int, mode: int = 0, reps: int = 3, clock: Optional[Callable] = None, **kw) -> "ShardedAggregator":
    """``exchange="auto"`` decided by MEASUREMENT (VERDICT r5 task 8): with more than one rank both forms are built -- the block
    all-gather and the halo exchange -- each runs ``reps`` whole aggregation steps at width ``dim`` on its already-built buffers
    (one untimed step first), the slowest rank's time counts (all-reduce MAX), and the faster form is kept; the other is dropped.
    The byte rule of ``ShardedAggregator(exchange="auto")`` cannot see that the halo form pays a send-side row gather and five
    library calls per step (config 5's rank shape: halo kernels 25-33 % slower at 0.46 x the bytes); it stays the rule for one
    rank and for emulated ranks, where there is no wire to time.  The result carries ``exchange_timed`` =
    {"allgather_ms", "halo_ms", "chosen", "reps", "dim"} (None when nothing was timed).  ``clock``: seconds-returning callable
    that has synchronised the device (tests inject one); default: perf_counter around device synchronisation."""
    import time
    group = kw.get("group")
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world <= 1 or kw.get("emulate") is not None:
        agg = ShardedAggregator(local_row_pointers, column_index, bounds, partSize, exchange="auto", **kw)
        agg.exchange_timed = None
        return agg

    def now(dev):
        if clock is not None:
            return clock()
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        return time.perf_counter()
    timed, built = {}, {}
    for form in ("allgather", "halo"):
        agg = ShardedAggregator(local_row_pointers, column_index, bounds, partSize, exchange=form, **kw)
        x = torch.ones(agg.n_local, int(dim), device=agg.device)
        deg = torch.ones(agg.n_local, device=agg.device) if mode == 1 else None
        agg.aggregate(x, mode, degrees_local=deg)                      # buffers, plans, communicator warm-up
        dist.barrier(group=group)
        t0 = now(agg.device)
        for _ in range(max(1, int(reps))):
            agg.aggregate(x, mode, degrees_local=deg)
        ms = (now(agg.device) - t0) * 1e3 / max(1, int(reps))
        t = torch.tensor([ms], dtype=torch.float64, device=agg._comm_device())
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)         # a step ends when its slowest rank does
        timed[form], built[form] = float(t.item()), agg
    chosen = "halo" if timed["halo"] < timed["allgather"] else "allgather"   # (the same numbers on every rank: the same choice)
    keep = built.pop(chosen)
    for other in built.values():
        other.release()
    keep.exchange_timed = {"allgather_ms": timed["allgather"], "halo_ms": timed["halo"], "chosen": chosen, "reps": int(reps),
                           "dim": int(dim)}
    return keep


def set_trace(sink: Optional[list]) -> None:
    """Debug / strict-test hook: while a list is installed, every intermediate of the sharded layers' forward and
    backward (dense update, aggregation input and output, partial and reduced weight gradient) is appended to it as
    (name, clone, HIP stream handle, thread id).  The clones are enqueued on the step's own stream: no synchronisation is
    added.  None switches it off (the default: the product never traces)."""
    global _trace_sink
    _trace_sink = sink


def _tr(name: str, t: Optional[torch.Tensor]) -> None:
    if _trace_sink is not None and t is not None:
        import threading
        stream = torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else -1
        _trace_sink.append((name, t.detach().clone(), stream, threading.get_ident()))


def _xtg(X: torch.Tensor, G: torch.Tensor) -> torch.Tensor:
    """X^T G (weight gradient): libgnna's MFMA kernel on the GPU, torch.mm for the CPU/gloo tests."""
    if X.is_cuda:
        from . import _lib
        return _lib.xtg(X, G)
    return torch.mm(X.t(), G)


def _all_reduce_sum(t: torch.Tensor, group, force: bool = False) -> torch.Tensor:
    if dist.is_initialized() and (dist.get_world_size(group) > 1 or force):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


class ShardedGCNFunction(torch.autograd.Function):
    """Y_local = Â[local rows, :] (X W):  update -> all-gather -> aggregate."""

    @staticmethod
    def forward(ctx, X_local, weight, agg: "ShardedAggregator", degrees_local):
        ctx.save_for_backward(X_local, weight)
        ctx.agg, ctx.deg = agg, degrees_local
        XW = torch.mm(X_local, weight)
        _tr("gcn.fwd.XW", XW)
        Y = agg.gcn(XW, degrees_local)
        _tr("gcn.fwd.Y", Y)
        return Y

    @staticmethod
    def backward(ctx, d_output):
        X_local, weight = ctx.saved_tensors
        d_output = d_output.contiguous()
        _tr("gcn.bwd.dY", d_output)
        G = ctx.agg.gcn(d_output, ctx.deg)                       # Â dY, rows of this rank
        _tr("gcn.bwd.G", G)
        d_input = torch.mm(G, weight.t()) if ctx.needs_input_grad[0] else None
        _tr("gcn.bwd.dX", d_input)
        d_weight = _xtg(X_local, G)
        _tr("gcn.bwd.dW.partial", d_weight)
        d_weight = _all_reduce_sum(d_weight, ctx.agg.group, ctx.agg.force_collectives)
        _tr("gcn.bwd.dW", d_weight)
        return d_input, d_weight, None, None


class ShardedGINFunction(torch.autograd.Function):
    """Y_local = (eps A[local rows, :] X) W:  all-gather -> aggregate -> update."""

    @staticmethod
    def forward(ctx, X_local, weight, agg: "ShardedAggregator", epsilon):
        _tr("gin.fwd.X", X_local)
        T = agg.gin(X_local, epsilon)
        _tr("gin.fwd.T", T)
        ctx.save_for_backward(T, weight)
        ctx.agg, ctx.epsilon = agg, epsilon
        return torch.mm(T, weight)

    @staticmethod
    def backward(ctx, d_output):
        T, weight = ctx.saved_tensors
        d_output = d_output.contiguous()
        _tr("gin.bwd.dY", d_output)
        d_weight = _xtg(T, d_output)
        _tr("gin.bwd.dW.partial", d_weight)
        d_weight = _all_reduce_sum(d_weight, ctx.agg.group, ctx.agg.force_collectives)
        _tr("gin.bwd.dW", d_weight)
        d_input = None
        if ctx.needs_input_grad[0]:
            M = torch.mm(d_output, weight.t())
            _tr("gin.bwd.dYWt", M)
            d_input = ctx.agg.gin(M, ctx.epsilon)
            _tr("gin.bwd.dX", d_input)
        return d_input, d_weight, None, None


class _ShardedConv(torch.nn.Module):
    """Weights are replicated: drawn like ops._NeighborConv, then broadcast from rank 0."""

    def __init__(self, input_dim: int, output_dim: int, agg: ShardedAggregator, device=None):
        super().__init__()
        self.agg = agg
        bound = 1.0 / (output_dim ** 0.5)
        w = torch.empty(input_dim, output_dim, device=device or agg.device).uniform_(-bound, bound)
        if dist.is_initialized() and (dist.get_world_size(agg.group) > 1 or agg.force_collectives):
            dist.broadcast(w, src=dist.get_global_rank(agg.group, 0) if agg.group is not None else 0,
                           group=agg.group)
        self.weights = torch.nn.Parameter(w)


class ShardedGCNConv(_ShardedConv):
    def forward(self, X_local: torch.Tensor, degrees_local: torch.Tensor) -> torch.Tensor:
        return ShardedGCNFunction.apply(X_local, self.weights, self.agg, degrees_local)


class ShardedGINConv(_ShardedConv):
    eplison = 0.5                                               # reference spelling, gnn_conv.py:132

    def forward(self, X_local: torch.Tensor) -> torch.Tensor:
        return ShardedGINFunction.apply(X_local, self.weights, self.agg, self.eplison)
