/*
 * gnna.h -- C ABI of libgnna.so, the MI355X (gfx950) neighbor-group aggregation runtime.
 *
 * This is the drop-in boundary for GNNAdvisor's aggregation hot path.  Every entry point
 * replaces one symbol the reference's extension binds (paths relative to
 * GNNAdvisor/GNNConv/ in the reference):
 *
 *   gnna_sag_f32            <- SAG_cuda                  GNNAdvisor_kernel.cu:110-184 (+ kernel :186-259)
 *   gnna_agg_gcn_f32        <- the aggregation half of spmm_forward_cuda   .cu:282-322 (+ kernel :324-415)
 *                              and of spmm_backward_cuda                   .cu:436-470 (+ kernel :478-552)
 *   gnna_agg_gin_f32        <- the aggregation half of spmm_forward_cuda_gin  .cu:575-603 (+ kernel :620-689)
 *                              and of spmm_backward_cuda_gin               .cu:712-744 (+ kernel :749-814)
 *   gnna_count_parts /
 *   gnna_build_part_i32     <- build_part                GNNAdvisor.cpp:210-251
 *
 * The dense update (torch::mm at .cu:280,472,473,605,710,711) stays with the caller's
 * BLAS exactly as in the reference; the pybind module `GNNAdvisor` shipped in
 * gnnadvisor_osdi21_amd/csrc/gnna_torch.cpp composes both and re-exports the reference's
 * six Python-visible functions (GNNAdvisor.cpp:253-263) with identical signatures.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / STL types.
 *   - feature matrices: float32, row-major, contiguous, `dim` floats per row
 *     (rows need only 4-byte alignment; 16-byte aligned bases and dim % 4 == 0 take the
 *     vectorised path).
 *   - index arrays: int32 (packed_accessor32<int,1> in the reference).
 *   - all pointers passed to gnna_sag/agg_* are DEVICE pointers; build_part pointers are HOST.
 *   - inputs are borrowed and never written; `out` is fully overwritten (the reference
 *     returns a fresh zeros_like + accumulate, .cu:121).
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls only
 *     enqueue work.  One exception: the first aggregation on a partition the library has not
 *     seen yet (and that is large enough to be worth slicing) runs a counting pass over the
 *     column ids and synchronises `stream` once to read its statistics; never while the stream
 *     is being captured.  Library scratch is allocated on first use and never grown during capture
 *     (GNNA_ERR_UNSUPPORTED instead): warm a path up once before capturing it -- or call
 *     gnna_prepare_graph() once per graph: after it no aggregation on that graph synchronises,
 *     allocates or frees, and a captured call takes the same schedule as an eager one.  Captured calls keep a block of
 *     per-call device scratch (step counters, lists) of their own for good, so graphs may be replayed concurrently; the
 *     library holds 160 such blocks per device -- graphs whose calls were captured after those are used up share a ring and
 *     must not be replayed concurrently with each other.
 *   - return value: GNNA_OK or a negative gnna_status; gnna_last_error() gives the
 *     message for the calling thread.  (The reference printf()s and exit(-1)s on launch
 *     failure, .cu:177-181; this library reports instead.)
 *   - partSize / dimWorker / warpPerBlock keep the reference's argument positions and
 *     are validated (> 0).  On CDNA4 they are scheduling hints: a 64-lane wavefront
 *     always covers the whole feature row, so dimWorker cannot drop dimensions the way
 *     dimWorker < 32 lanes strides them in the reference (.cu:246).
 */
#ifndef GNNA_H_
#define GNNA_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNNA_VERSION 600 /* 0.6.0: gnna_tuning.ids_check_every (full hash of the ids behind a packed copy), gnna_device_cus, gnna_host_threads; 0.5.0: gnna_tuning opens with struct_size (checked by gnna_set_tuning, which now returns a status), gnna_build_id; 0.4.1: gnna_sddmm_ld_f32 (leading dimensions for both SDDMM sides); 0.4.0: gnna_agg_ld_f32 (leading dimensions, ReLU epilogue), gnna_forget_graph, chunk-walk kernel retired; 0.3.1: gnna_tuning grew (pack_ids); 0.3.0: sweep, sweep_slack, graph lifecycle, 64-bit CSR builder */
#define GNNA_API __attribute__((visibility("default")))

typedef enum gnna_status {
    GNNA_OK = 0,
    GNNA_ERR_INVALID_ARGUMENT = -1,
    GNNA_ERR_HIP = -2,
    GNNA_ERR_UNSUPPORTED = -3
} gnna_status;

GNNA_API int gnna_version(void);
/* "<major>.<minor>.<patch>+<16 hex digits>": the digits are the SHA-256 prefix of the library's sources (every file of
 * csrc/ and include/ that goes into libgnna.so and the GNNAdvisor module) as the build script hashed them -- a run can
 * prove that the binary it loaded was built from the sources beside it (gnnadvisor_osdi21_amd/build.py: source_hash). */
GNNA_API const char *gnna_build_id(void);
GNNA_API const char *gnna_last_error(void);

/* What the planners size their work for (param.py:4-164 reads the SM count and shared memory of "a GPU" from constants;
 * here the Decider asks): compute units of the calling thread's current device (256 on MI355X; 0 when no device is visible --
 * never an error) and the host threads the native builders / the renumbering may use (the CPUs this process is allowed:
 * affinity mask, cgroup quota on the way up from its own cgroup, GNNA_HOST_THREADS overrides; at most 64). */
GNNA_API int gnna_device_cus(void);
GNNA_API int gnna_host_threads(void);

/* ---- partitioner (host) --------------------------------------------------------------
 * Neighbor-group partition of a CSR: row i with degree d_i yields ceil(d_i / partSize)
 * groups of at most partSize consecutive neighbors; group p covers edge offsets
 * [partPtr[p], partPtr[p+1]) and belongs to row part2Node[p].
 * Replaces build_part (GNNAdvisor.cpp:210-251).  Deliberate divergences:
 * exact int32 storage instead of float32, and partPtr[P] = indptr[N] is always written.
 */
GNNA_API int64_t gnna_count_parts(int partSize, const int32_t *indptr, int64_t num_nodes);
GNNA_API int gnna_build_part_i32(int partSize, const int32_t *indptr, int64_t num_nodes,
                        int32_t *partPtr /* [num_parts + 1] */,
                        int32_t *part2Node /* [num_parts] */, int64_t num_parts);

/* ---- graph inputs (host) --------------------------------------------------------------
 * Native counterparts of the reference loader's CSR construction
 * (GNNAdvisor/dataset.py:99-122) and of the renumbering hook (rabbit.reorder,
 * rabbit_module/src/reorder.cpp:235-295).  All pointers are HOST pointers.
 */

/* Edge list -> CSR with scipy coo->csr semantics (dataset.py:108-118): duplicate edges
 * merged, column indices sorted per row.  column_index must have room for num_edges
 * entries; the return value is nnz (>= 0) or a negative gnna_status. */
GNNA_API int64_t gnna_csr_from_edges_i32(const int32_t *src, const int32_t *dst, int64_t num_edges,
                                int64_t num_nodes, int32_t *row_pointers /* [num_nodes + 1] */,
                                int32_t *column_index /* [num_edges] */);

/* Sharded ingestion (multi-GPU, SURVEY 8e "index width"; extends dataset.py:99-122, which builds one int32 CSR):
 * the edge LIST may hold more than 2^31 entries (papers100M symmetrised: 3.2e9) and is counted with 64-bit
 * arithmetic; every rank then builds only the CSR rows of its own destination range, whose edge count must fit
 * int32 (GNNA_ERR_UNSUPPORTED otherwise: use more ranks).
 *   gnna_row_counts_i64: counts[r] += number of list entries with rows[e] == r (accumulates, so a list can be fed
 *     in pieces; raw entries -- duplicates are merged later, per shard);
 *   gnna_row_splits_i64: global 64-bit row pointers of those counts (row_pointers may be NULL) and `world` + 1
 *     row bounds that cut the rows into contiguous blocks of about equal edge count;
 *   gnna_csr_from_edges_range_i32: the CSR of rows [row_lo, row_hi) only -- local int32 row_pointers
 *     [row_hi - row_lo + 1] rebased to 0, GLOBAL column ids, duplicates merged, columns sorted per row (the
 *     semantics of gnna_csr_from_edges_i32); returns the shard's nnz (<= capacity) or a negative gnna_status. */
GNNA_API int gnna_row_counts_i64(const int32_t *rows, int64_t num_edges, int64_t num_nodes, int64_t *counts /* [num_nodes] */);
GNNA_API int gnna_row_splits_i64(const int64_t *counts, int64_t num_nodes, int world, int64_t *bounds /* [world + 1] */,
                                 int64_t *row_pointers /* [num_nodes + 1] or NULL */);
GNNA_API int64_t gnna_csr_from_edges_range_i32(const int32_t *src, const int32_t *dst, int64_t num_edges,
                                               int64_t num_nodes, int64_t row_lo, int64_t row_hi,
                                               int32_t *row_pointers, int32_t *column_index, int64_t capacity);

/* degrees[i] = sqrt(max(row_pointers[i+1] - row_pointers[i], 1))  (dataset.py:11-18,121-122) */
GNNA_API int gnna_degrees_f32(const int32_t *row_pointers, int64_t num_nodes, float *degrees);

/* avg_edge_span = mean |src - dst| over the raw edge list (dataset.py:100; Decider input) */
GNNA_API int gnna_edge_span(const int32_t *src, const int32_t *dst, int64_t num_edges, double *avg_edge_span);

/* Locality renumbering: writes new_id[old_id] (a permutation of 0..num_nodes-1) computed by
 * a deterministic reverse Cuthill-McKee sweep over the symmetrised graph (components in
 * order of their lowest-degree seed).  Same role as rabbit.reorder -- shrink the average
 * edge span -- but a different, single-threaded, reproducible algorithm. */
GNNA_API int gnna_reorder_rcm_i32(const int32_t *src, const int32_t *dst, int64_t num_edges,
                         int64_t num_nodes, int32_t *new_id /* [num_nodes] */);

/* Locality renumbering by communities: writes new_id[old_id] (a permutation).  On the symmetrised graph
 * (multi-threaded, deterministic): the triangle-supported backbone of the graph (edges whose end points share
 * neighbours), a breadth-first order over it in which a node is discovered once several of its backbone neighbours
 * have been walked (unfolded when the walk runs on two fronts), then median sweeps -- the role of rabbit.reorder
 * (community-based Rabbit Order, rabbit_module/src/reorder.cpp:235-295) with a different, reproducible algorithm
 * (gnna_reorder.cpp).  Node ids are int32; the edge list may hold any number of entries (64-bit
 * offsets inside: papers100M symmetrised has 3.2e9). */
GNNA_API int gnna_reorder_community_i32(const int32_t *src, const int32_t *dst, int64_t num_edges,
                               int64_t num_nodes, int32_t *new_id /* [num_nodes] */);

/* The same renumbering from the loader's CSR (sorted, duplicate-free rows; checked): a symmetric CSR is used as the adjacency
 * as it stands -- no second counting sort (1.2 of 4.6 s at 1.1e8 edges); a directed one goes through the edge-list form.
 * The permutation is the one gnna_reorder_community_i32 returns for the same graph. */
GNNA_API int gnna_reorder_community_csr_i32(const int32_t *row_pointers, const int32_t *column_index, int64_t num_nodes,
                                            int32_t *new_id /* [num_nodes] */);

/* Applying a renumbering to what the loader holds (dataset.py:147-172 relabels the edge list, then rebuilds CSR and degrees from
 * it): the edge list in place (+ the new mean |src - dst|, the Decider's statistic), and the relabelled graph's CSR straight from
 * the old CSR -- row new_id[u] = the ids of row u mapped and sorted; no global sort (a permutation keeps rows duplicate-free). */
GNNA_API int gnna_relabel_edges_i32(int32_t *src, int32_t *dst, int64_t num_edges, const int32_t *new_id, int64_t num_nodes,
                                    double *avg_edge_span /* may be NULL */);
GNNA_API int gnna_relabel_csr_i32(const int32_t *row_pointers, const int32_t *column_index, int64_t num_nodes,
                                  const int32_t *new_id, int32_t *out_row_pointers /* [num_nodes + 1] */,
                                  int32_t *out_column_index /* [nnz] */);

/* ---- aggregation (device) ------------------------------------------------------------
 * out[i, :] = sum over groups p with part2Node[p] == i, over e in [part_pointers[p],
 *             part_pointers[p+1]):  coef(i, column_index[e]) * input[column_index[e], :]
 *   sag : coef = 1
 *   gcn : coef = degrees[i] * degrees[column_index[e]]     (a product, as in the reference)
 *   gin : coef = 1, the row sum is scaled by epsilon
 * row_pointers and (for sag) degrees are accepted for signature parity and unused, as
 * in the reference kernels.
 */
/* Limits of one call (GNNA_ERR_UNSUPPORTED beyond them): fewer than 2^29 destination rows, fewer than 2^31 edges
 * (int32 CSR), source matrices of any size (64-bit row offsets above 4 GiB). */
GNNA_API int gnna_sag_f32(const float *input, const int32_t *row_pointers, const int32_t *column_index,
                 const float *degrees, const int32_t *part_pointers, const int32_t *part2Node,
                 float *out, int64_t num_nodes, int dim, int64_t num_parts,
                 int partSize, int dimWorker, int warpPerBlock, void *stream);

GNNA_API int gnna_agg_gcn_f32(const float *input, const int32_t *row_pointers, const int32_t *column_index,
                     const float *degrees, const int32_t *part_pointers, const int32_t *part2Node,
                     float *out, int64_t num_nodes, int dim, int64_t num_parts,
                     int partSize, int dimWorker, int warpPerBlock, void *stream);

GNNA_API int gnna_agg_gin_f32(const float *input, const int32_t *row_pointers, const int32_t *column_index,
                     float epsilon, const int32_t *part_pointers, const int32_t *part2Node,
                     float *out, int64_t num_nodes, int dim, int64_t num_parts,
                     int partSize, int dimWorker, int warpPerBlock, void *stream);

/* Rectangular form for a destination-range shard (multi-GPU; no counterpart in
 * the single-GPU reference): `out` has num_out_rows rows (the shard's destination nodes),
 * `input` has num_in_rows rows (all source nodes, e.g. the all-gathered feature matrix) and
 * column_index holds ids into `input`.  mode: 0 = sag, 1 = gcn, 2 = gin.  For gcn,
 * degrees_out is indexed by destination row and degrees_in by source row.
 * accumulate != 0 adds into the existing contents of `out` instead of overwriting it (used to
 * aggregate local-source edges while the all-gather of remote features is still in flight,
 * then add the remote-source part).
 */
GNNA_API int gnna_agg_rect_f32(int mode, const float *input, int64_t num_in_rows,
                      const int32_t *column_index, const float *degrees_out, const float *degrees_in,
                      float epsilon, const int32_t *part_pointers, const int32_t *part2Node,
                      float *out, int64_t num_out_rows, int dim, int64_t num_parts, int partSize,
                      int accumulate, void *stream);

/* General form: gnna_agg_rect_f32 with leading dimensions and an epilogue.
 *   ld_in / ld_out: floats between the starts of consecutive rows of `input` / `out` (>= dim; == dim: the contiguous
 *     layout every other entry point assumes -- the reference insists on contiguous tensors, GNNAdvisor.cpp:71-73).
 *     With them a producer can write the layout the gather likes best itself -- rows of 17..64 floats that start at
 *     multiples of 2 x ceil(4 dim / 128) x 128 bytes (stride 64 / 128 / 128 floats for dim 32 / 41 / 64), rows padded
 *     to a 128-byte-line-friendly stride (41 -> 48) -- and the library then gathers from `input` directly instead of
 *     staging a copy per call; a column block of a wider matrix (dim = 64, ld = 256) can be aggregated in place, and
 *     the result can land in a slice of a wider buffer (concatenated layer outputs).
 *   flags: GNNA_ACCUMULATE     add into the existing contents of `out` (the `accumulate` of gnna_agg_rect_f32);
 *          GNNA_EPILOGUE_RELU  out = max(out, 0) after the aggregation (and the accumulate): fused into the store wherever a
 *                              row is written once (sweep kernel, owned rows of a single pass), a small pass over the rows
 *                              that several work items add to otherwise.  Reference call sites: F.relu(conv(...)) after
 *                              every layer, GNNA_main.py:151,166-169.
 * mode, degrees_out / degrees_in, epsilon as for gnna_agg_rect_f32. */
#define GNNA_ACCUMULATE 1u
#define GNNA_EPILOGUE_RELU 2u
GNNA_API int gnna_agg_ld_f32(int mode, const float *input, int64_t ld_in, int64_t num_in_rows,
                      const int32_t *column_index, const float *degrees_out, const float *degrees_in,
                      float epsilon, const int32_t *part_pointers, const int32_t *part2Node,
                      float *out, int64_t ld_out, int64_t num_out_rows, int dim, int64_t num_parts, int partSize,
                      unsigned flags, void *stream);

/* The row stride (in floats, >= dim) the library would stage `dim`-float source rows into before gathering them
 * `num_edges` times from `num_in_rows` rows -- dim itself when it would gather from the contiguous layout.  A producer
 * that writes its output with this leading dimension into a buffer aligned to (stride x 4) bytes (torch::mm into
 * buf[:, :dim] of a [N, stride] allocation) and calls gnna_agg_ld_f32 saves the library's staging copy per call. */
GNNA_API int64_t gnna_preferred_ld(int dim, int64_t num_in_rows, int64_t num_edges);

/* Windowed form of gnna_agg_rect_f32, for pipelining the aggregation with a chunked feature
 * exchange (multi-GPU: the all-gather is issued in `num_windows` pieces and each piece is
 * aggregated as soon as it has arrived).  The source rows are cut into `num_windows` (1..16) equal
 * windows of ceil(num_in_rows / num_windows) rows; this call aggregates the edges whose source
 * lies in windows [window_begin, window_end) and must only read those rows of `input`.
 * Contract: the calls of one aggregation cover every window exactly once, in increasing window order, with
 * identical other arguments; the call with window_begin == 0 overwrites `out` unless accumulate != 0, later calls
 * add.  The column ids of every neighbor-group must be in non-decreasing order (the loader's CSR; checked -- once per
 * graph, by the counting pass that builds the per-window counts -- and refused with GNNA_ERR_UNSUPPORTED otherwise:
 * a window call takes the id POSITIONS of its windows, which are its ids only when the ids are sorted).  Stateless
 * between the calls (0.4.0: runs on the streaming kernel; the per-run cursors of the earlier implementation are gone);
 * the first call of a sequence on a graph the library has not seen synchronises the stream once (not inside a stream
 * capture: GNNA_ERR_UNSUPPORTED there -- run the sequence once before capturing it).
 */
GNNA_API int gnna_agg_rect_windows_f32(int mode, const float *input, int64_t num_in_rows,
                      const int32_t *column_index, const float *degrees_out, const float *degrees_in,
                      float epsilon, const int32_t *part_pointers, const int32_t *part2Node,
                      float *out, int64_t num_out_rows, int dim, int64_t num_parts, int partSize,
                      int accumulate, int num_windows, int window_begin, int window_end, void *stream);

/* SDDMM over the same neighbor-group partition -- a build-defined extension: the reference
 * contains no SDDMM kernel, BASELINE.json's north star asks for one.
 *   edge_out[e] = < dst_feat[part2Node[p], :], src_feat[column_index[e], :] >
 * for every group p and every e in [part_pointers[p], part_pointers[p+1]); edge_out is indexed like
 * column_index.  dim >= 4.  Parity is defined by the dense formula only (no reference behaviour).
 */
GNNA_API int gnna_sddmm_f32(const float *dst_feat, const float *src_feat, const int32_t *column_index,
                   const int32_t *part_pointers, const int32_t *part2Node, float *edge_out,
                   int64_t num_out_rows, int64_t num_in_rows, int dim, int64_t num_parts, int partSize,
                   void *stream);

/* The same with leading dimensions (floats, >= dim) for both feature matrices: row r of dst_feat starts at
 * dst_feat + r * ld_dst, row r of src_feat at src_feat + r * ld_src -- a column block of a wider matrix (one attention
 * head of [N, heads * dim]) or the gapped layout gnna_preferred_ld names for the gathered side (src_feat). */
GNNA_API int gnna_sddmm_ld_f32(const float *dst_feat, int64_t ld_dst, const float *src_feat, int64_t ld_src,
                   const int32_t *column_index, const int32_t *part_pointers, const int32_t *part2Node, float *edge_out,
                   int64_t num_out_rows, int64_t num_in_rows, int dim, int64_t num_parts, int partSize,
                   void *stream);

/* Weight gradient of the dense update: dW[K, N] = X^T[K, M] * G[M, N], X = [num_rows, K] and
 * G = [num_rows, N] row-major fp32 (reference: torch::mm(X.transpose(0,1), d_input_prime),
 * GNNAdvisor_kernel.cu:473 and :710).  A reduction over the node dimension on the MFMA units
 * (v_mfma_f32_16x16x4_f32, fp32 in / fp32 accumulate); deterministic (no global atomics).
 * dW is overwritten.  Uses library scratch of the stream (first use allocates).
 */
GNNA_API int gnna_xtg_f32(const float *X, const float *G, float *dW, int64_t num_rows, int K, int N,
                 void *stream);

/* ---- scheduling knobs (not part of the reference API; used by the tuner and bench) ----
 * Any field <= 0 (or < 0 where 0 is meaningful) keeps the built-in choice.
 */
typedef struct gnna_tuning {
    int struct_size;      /* sizeof(gnna_tuning) of the caller's header: gnna_set_tuning refuses any other value (fields came
                             and went between 0.3 and 0.4 without a check; a caller built against another layout now gets
                             GNNA_ERR_INVALID_ARGUMENT instead of silently shifted knobs).  gnna_get_tuning fills it in. */
    int groups_per_chunk; /* neighbor-groups a wavefront walks per work item (1..64)      */
    int loads_in_flight;  /* wave-wide row loads issued before the first add (4, 8, 16)   */
    int blocks_per_cu;    /* sweep kernel only: workgroups per CU (1: one 16-wavefront workgroup with all of the
                             CU's LDS, 2: two); the streaming kernel's grid is hardware scheduled            */
    int xcd_remap;        /* 1: consecutive work items stay on one XCD's L2; 0: off       */
    int trust_canonical;  /* 1: skip the partition validation pass (build_part output)    */
    int column_phases;    /* 1: single pass; 2..32: gather X in that many source-id ranges (cache-resident
                             slices, all in one launch); 0: automatic, from the library's own statistics of
                             the partition */
    int avg_degree;       /* hint: average edges per destination row (0 = unknown)           */
    int nonlocal_ids;     /* hint: 1 = source ids of a row are scattered over the whole id
                             range (no community ordering), 0 = unknown / locality-ordered  */
    int gcn_prescale;     /* gnna_agg_gcn_f32: 1 = scale the source rows by their degree norm
                             once (library workspace) and gather unweighted, 2 = per-edge
                             coefficients as the reference computes them, 0 = automatic
                             (pre-scale when a source row is gathered >= ~32 times)          */
    int pad_rows;         /* 1 = gather from a staged copy of the source rows whose row stride is
                             padded to a 128-byte-line-friendly size when the width calls for
                             it (e.g. 41 -> 48, 56 -> 64 floats), 2 = never, 0 = automatic (when
                             a source row is gathered >= ~32 times; rows of 17..64 floats then start at
                             multiples of twice their 128-byte lines -- stride 64 / 128 / 128 floats for
                             32 / 41 / 64 -- while the copy stays Infinity-Cache sized: 2-6 % off the
                             gather), > 2 = this stride in floats (experiments) */
    int zero_fill;        /* what the prologue clears before a single pass of the streaming kernel that overwrites
                             `out`: 1 = only the rows that pass does not store (rows without edges, rows shared by
                             two work items), 2 = the whole output, 0 = automatic (1 once `out` is >= 32 MiB).
                             Every other schedule adds into `out` and always clears all of it */
    int sweep;            /* destination-blocked sweep kernel (gnna_sweep.hip): persistent wavefronts keep the partial
                             rows of their groups in LDS while every XCD walks the source slices in step, and write
                             each row once.  1 = wherever it applies (sliced schedule, unweighted or pre-scaled gather,
                             rows of >= 4 floats <= 128 wide, source matrix <= 4 GiB), 2 = never, 0 = automatic: where it
                             measures faster than the streaming kernel -- rows of 33..64 floats, a sliced schedule over a
                             square, Infinity-Cache-sized problem, long rows (>= 300 edges on average) and few enough of
                             them that a workgroup's share fits its LDS accumulators in two sets.  While it is selected by 1 two other knobs are read in its terms:
                             blocks_per_cu = workgroups per CU (1: one 16-wavefront workgroup with all of the CU's LDS,
                             else two), groups_per_chunk > 64 = 64 x (sets per workgroup) instead of the automatic count */
    int sweep_slack;      /* sweep kernel: how many slice steps a wavefront may run ahead of the slowest wavefront of
                             its XCD (soft barrier on a per-XCD counter, bounded spin: only locality depends on it).
                             0 = built-in, n > 0 = n steps, >= 1000 = no synchronisation at all */
    int deterministic;    /* 1 = bit-reproducible results of the streaming kernel for a canonical partition: the phases
                             of the sliced schedule run as separate launches in order, a row owned by one work item is
                             read-modify-written, and the partial sums of a row shared between work items are parked in
                             library scratch and added in work-item order by a small second kernel -- no float atomics,
                             so the association order of every output element is fixed.  Costs the per-phase launches
                             and the read-back of the rows; 0 = the default schedule (atomics: reproducible to fp32
                             rounding, exact where the sums are exactly representable) */
    int pack_ids;         /* prepared graphs (gnna_prepare_graph) only: keep, per phase count in use, a copy of the column
                             ids in the order the sliced schedule consumes them, so that a work item reads its ids once
                             and contiguously instead of one cache line per neighbor-group and phase (nnz x 4 bytes per
                             copy, up to 4 copies per graph).  0 = automatic (on for prepared graphs), 2 = off,
                             (every call compares 1024 samples of column_index with the copy's checksum and reads
                             column_index itself on a mismatch: a rewritten buffer is noticed, a few changed ids are not);
                             1 = on for EVERY graph, prepared or not: the caller promises that no column_index is
                             rewritten in place while the library holds a plan for it (the reference's own call sequence
                             never does; without the promise a stale copy would give wrong results, which is why it is
                             not the default) */
    int ids_check_every;  /* packed ids: every n-th aggregation that reads a packed copy also compares a 64-bit hash of ALL of
                             column_index and part_pointers (one read of both: ~0.1 ms per 114 M ids) with the hash taken when the
                             copy was made; on a mismatch that call and every later one read column_index itself (the copy is
                             marked on the device and never trusted again: gnna_forget_graph / gnna_prepare_graph make a new one).
                             Closes what the 2 x 1,024 samples leave open -- a few ids rewritten in place -- within n calls.
                             <= 0 = built-in (64: < 0.2 % of the aggregation time), 1 = every call (GNNA_DEBUG_FULL_CHECKSUM=1 sets it:
                             the test suite), n > 1 = every n-th, >= 2^30 = never.  Not inside a stream capture (samples only there) */
    int wide_blocks;      /* wide rows in 64-float column blocks (one call per block, leading dimensions): 0 = automatic (rows of
                             >= 100 floats of long-row graphs, >= 192 otherwise, when every source row is gathered >= ~32 times
                             and the matrix is Infinity-Cache sized), 1 = whenever dim >= 72, 2 = never */
} gnna_tuning;

GNNA_API int gnna_set_tuning(const gnna_tuning *t); /* NULL restores the defaults; GNNA_ERR_INVALID_ARGUMENT on a struct_size mismatch */
GNNA_API void gnna_get_tuning(gnna_tuning *t);

/* Per-graph form of the two hints: remembers (avg_degree, nonlocal_ids) for the graph whose
 * column_index array starts at this device address; aggregation calls on that array use them
 * instead of the process-wide gnna_tuning values.  avg_degree <= 0 forgets the graph,
 * column_index == NULL forgets all graphs.  At most 64 graphs are remembered (least recently used
 * is replaced).  A stale entry can only cost performance, never correctness. */
GNNA_API int gnna_set_graph_hints(const int32_t *column_index, int avg_degree, int nonlocal_ids);

/* Measured schedule: aggregations of `dim`-wide features on the graph whose column_index array
 * starts at this device address use `column_phases` (1..32) column phases instead of the rule
 * based on the hints; 0 removes the entry.  Up to 8 widths per graph.  An explicit process-wide
 * gnna_tuning.column_phases >= 1 still wins.  (decider.inputProperty.calibrate() measures and
 * registers these.) */
GNNA_API int gnna_set_graph_phases(const int32_t *column_index, int dim, int column_phases);

/* ---- graph lifecycle (optional) ----------------------------------------------------------
 * gnna_prepare_graph does, up front and on `stream`, everything an aggregation call would otherwise do at first
 * sight of a partition: the counting pass over the column ids, the read-back of its statistics (ONE
 * synchronisation of `stream`, here instead of inside the first aggregation), the choice of the phase count for
 * every feature width in dims[0 .. num_dims) (written to phases_out when not NULL) and the sizing of the stream's
 * scratch (row staging for widths that want it).  The plan is pinned: it is not subject to the library's
 * least-recently-used replacement and lives until gnna_release_graph.  After a successful prepare, aggregation
 * calls on (column_index, part_pointers, part2Node) gathering from `num_in_rows` source rows never synchronise,
 * allocate or free, and a call inside a stream capture takes the sliced schedule like an eager call.
 * num_out_rows: destination rows (rows of `out`); partSize as passed to the aggregation calls.
 * All pointers are DEVICE pointers and must stay valid (and unchanged in content) until the release.
 * gnna_release_graph forgets every plan, hint and measured schedule keyed by this column_index address
 * (NULL: all graphs); it waits for the device before freeing. */
GNNA_API int gnna_prepare_graph(const int32_t *column_index, const int32_t *part_pointers, const int32_t *part2Node,
                                int64_t num_parts, int64_t num_in_rows, int64_t num_out_rows, int partSize,
                                const int *dims, int num_dims, int *phases_out, void *stream);
GNNA_API int gnna_release_graph(const int32_t *column_index);
/* The same for callers that cannot choose their moment -- finalizers, garbage collectors, another thread while a stream is
 * being captured: the graph's plans, hints and schedules are unlinked at once (no later call finds them), but nothing is
 * freed and nothing synchronises here; the device buffers go at the next gnna_prepare_graph / gnna_release_graph, or when
 * the library next allocates a plan, and never while another thread's aggregation call is between looking a plan up
 * and enqueueing its kernels. */
GNNA_API int gnna_forget_graph(const int32_t *column_index);
/* gnna_forget_graph without touching what the CALLER registered for the graph: the plans and packed id copies go (deferred, as
 * above), the hints (gnna_set_graph_hints) and measured schedules (gnna_set_graph_phases) stay.  For a layer that pinned plans
 * on the caller's behalf and has to take them back -- the GNNAdvisor module's automatic lifecycle -- while the graph lives on. */
GNNA_API int gnna_forget_plans(const int32_t *column_index);

/* Events of the library's launch path since load (for tests and monitoring):
 *   [0] counting passes run, [1] stream / device synchronisations inside aggregation calls, [2] hipFree and
 *   [3] hipMalloc inside aggregation calls, [4] counting passes skipped by the back-off for partitions that are
 *   never seen twice, [5] launches of the sweep kernel, [6] packed-id copies built, [7] aggregation launches that
 *   read packed ids. */
GNNA_API void gnna_runtime_counters(int64_t out[8]);
/* The same list, open-ended: fills out[0 .. min(capacity, count)) and returns the number of counters this library keeps.
 *   [8] full hashes of the ids behind a packed copy run at a launch (gnna_tuning.ids_check_every). */
GNNA_API int gnna_runtime_counters_ex(int64_t *out, int capacity);

/* Number of column phases the calling thread's most recent aggregation call used (>= 1). */
GNNA_API int gnna_last_num_phases(void);
/* Number of aggregation-kernel launches that call issued (the sliced schedule of the streaming kernel is one
 * launch whatever its number of phases; the chunk-walk kernel launches once per phase). */
GNNA_API int gnna_last_num_launches(void);

/* ---- kernel timing (HIP events on the caller's stream; used by bench.py) ---------------
 * Between gnna_profile_begin() and gnna_profile_end() every aggregation call records HIP
 * events around its prologue kernel and around its main kernel on the stream it was
 * given.  gnna_profile_end() synchronises those events and returns the average
 * durations in milliseconds over the recorded calls (at most max_calls are recorded).
 */
GNNA_API int gnna_profile_begin(int max_calls);
GNNA_API int gnna_profile_end(double *avg_main_ms, double *avg_prologue_ms, int *num_calls);

#ifdef __cplusplus
}
#endif
#endif /* GNNA_H_ */
