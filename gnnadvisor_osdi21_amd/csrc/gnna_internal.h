// gnna_internal.h -- shared by the translation units of libgnna.so (not installed).
#ifndef GNNA_INTERNAL_H_
#define GNNA_INTERNAL_H_

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstddef>
#include <cstdint>
#include <map>
#include <utility>

#include "gnna.h"

namespace gnna {

// Records a formatted message for gnna_last_error() on this thread and returns `code`.
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// Host worker threads the library's CPU passes (CSR builder, renumbering) should start: the hardware threads, capped by the
// CPUs the container is actually GRANTED (cgroup v2 cpu.max / v1 cfs quota) and by GNNA_HOST_THREADS.  The GPU boxes show
// 256 hardware threads and grant 16 CPUs: 64 threads there spend most of every scheduling period throttled (round 5: the
// community renumbering of the Reddit-like graph took 90 s with 64 threads on a 16-CPU quota).
int host_thread_budget(int cap);

// Per-graph hints registered with gnna_set_graph_hints(), keyed by the column_index pointer of the
// call; overrides tune->avg_degree / nonlocal_ids when there is an entry, and tune->column_phases when a
// measured schedule for this feature width was registered with gnna_set_graph_phases().  (gnna_host.cpp)
void apply_graph_hints(const void *column_index, int dim, gnna_tuning *tune);

// ---- per-device runtime state (gnna_runtime.hip) -------------------------------------------------
struct Workspace {
    void *ptr = nullptr;
    size_t bytes = 0;
    bool captured = false;     // handed to a call that was being captured: a graph points at it, it is never freed
};
struct DeviceState {
    std::atomic<bool> init{false};
    int num_cus = 256;
    int32_t *flags = nullptr;  // two arrays of kFlagSlots ints, zero-initialised: [slot] "partition is not canonical",
                               // [kFlagSlots + slot] "the packed ids are stale" -- each holds the sequence number of the call
                               // that raised it; slot = the call's block (call_block_of)
    unsigned long long *gap_lists = nullptr;   // kCallBlocks lists of kGapWords words (call_block_of)
    uint32_t *sweep_sync = nullptr;  // kCallBlocks counter blocks for the sweep kernel's soft barrier (call_block_of)
    std::map<hipStream_t, int> stream_block;   // streams that own one of the first kStreamBlocks call blocks
    int captured_blocks = 0;                   // call blocks handed to captured calls so far (never returned)
    std::map<std::pair<hipStream_t, int>, Workspace> ws;  // per stream: slot 1 staged / pre-scaled X, slot 2 partial rows of the
                                                          // deterministic schedule / slabs of the weight-gradient kernel
};
constexpr int kFlagSlots = 1024;
// per-call lists of long runs of rows without edges that the sparse prologue leaves to a grid-wide pass
constexpr int kGapEntries = 63, kGapWords = 2 + 2 * kGapEntries;   // word 0: count, pairs (first row, rows)
// Per-call device scratch that is NOT tagged with the call's sequence number (the gap list of the sparse prologue, the sweep
// kernel's step counters and ReLU list) lives in "call blocks".  Work on one stream is ordered, so a stream needs one block:
//   * the first kStreamBlocks streams that call the library get a block of their own (no allocation);
//   * a call that is being CAPTURED into a graph runs whenever and wherever its graph is replayed -- not in the order of the
//     capture stream (torch.cuda.graph captures every graph on one shared side stream) and possibly next to a replay of another
//     graph: each of the first kCapturedBlocks captured calls of a device gets a block of its own FOR GOOD (the graph may be
//     replayed at any time; round 6, ADVICE r5 -- before, they shared the ring below by sequence number);
//   * calls on further streams, and captured calls beyond those, share a ring of the remaining blocks by call count: graphs
//     captured after the first kCapturedBlocks calls must not be replayed concurrently with each other (include/gnna.h).
// (A ring alone, shared by all streams, lets a call on one stream clear the block of a call that is still running on another.)
constexpr int kStreamBlocks = 64, kCapturedBlocks = 160, kCallBlocks = 256;
static_assert(kCallBlocks <= kFlagSlots && kStreamBlocks + kCapturedBlocks < kCallBlocks, "call blocks");

// State of the current device (lazily created: CU count, flag ring).
int get_device_state(DeviceState **out);
// Grow-only scratch buffer `slot` of `stream`.
int get_workspace(DeviceState *ds, hipStream_t stream, int slot, size_t bytes, void **out);
// Fresh non-zero sequence number of an aggregation call on `stream` and its slot among the flags (the call block's index).
int32_t next_call_seq(DeviceState *ds, hipStream_t stream, int32_t **flag_slot);
// Index of the call block (see kCallBlocks) of a call with sequence number `seq` on `stream`.
int call_block_of(DeviceState *ds, hipStream_t stream, int32_t seq);

// ---- streaming kernel + sliced schedule (gnna_stream.hip) ----------------------------------------------
constexpr int kMaxSlices = 32;    // fine source slices of a slice plan (one byte of cumulative count per group and boundary)
constexpr int kSliceLevels = 5;   // cells[l]: non-empty (group, slice) cells when the fine slices are merged into kMaxSlices >> l
struct SlicePlanStats {
    bool valid = false;
    double cells[kSliceLevels] = {0, 0, 0, 0, 0};
    double edges = 0, groups = 0;
    double span = 0;                 // sum over the edges of |column id - destination row|
    double unsorted = 0;             // neighbor-groups whose column ids are not in non-decreasing order
    double near[24] = {0};           // near[k]: edges with |column id - destination row| < 256 * 2^(k / 2)
};
struct SlicePlan {
    const uint8_t *cnt = nullptr;    // [S - 1][P] cumulative slice counts (phase-major), null: no plan right now
    int S = 0;                       // fine slices of the plan
    uint32_t slice_rows = 0;         // source rows per fine slice
    SlicePlanStats stats;
    void *handle = nullptr;          // the plan itself (get_packed_ids)
    bool pinned = false;             // made by gnna_prepare_graph (the caller promised not to change the graph)
};
// Fine slicing of `num_in_rows` source rows: S = kMaxSlices slices of ceil(num_in_rows / S) rows.
inline uint32_t slice_rows_for(int64_t num_in_rows)
{
    const int64_t r = (num_in_rows + kMaxSlices - 1) / kMaxSlices;
    return (uint32_t)(r < 1 ? 1 : r);
}
// Slice plan of the partition for gathers from `num_in_rows` source rows (library cache keyed by the device
// addresses; a miss enqueues the counting kernel on `stream`, and with want_stats synchronises it once to read
// the statistics).  out->cnt stays null when the plan cannot be built right now (stream capture, or the
// back-off for partitions that are never seen twice).  A plan made by gnna_prepare_graph is never evicted.
int get_slice_plan(DeviceState *ds, hipStream_t stream, const int32_t *column_index, const int32_t *part_pointers,
                   const int32_t *part2Node, int64_t num_parts, int64_t num_in_rows, bool want_stats, bool pin,
                   SlicePlan *out, uint32_t window_rows = 0);
void drop_slice_plans();
// Packed column ids of a plan for (B phases, G groups per chunk): see gnna_stream.hip.  *ids == null: none.  The
// caller decides whether the plan may have them (pinned, or gnna_tuning.pack_ids = 1).
int get_packed_ids(DeviceState *ds, hipStream_t stream, void *plan_handle, int B, int G, bool may_build, bool force,
                   const int32_t **ids, const uint32_t **item_off, const unsigned long long **checksum = nullptr,
                   int64_t *num_ids = nullptr, int32_t *stale_flag = nullptr, int32_t seq = 0, int check_every = 0);
// One-wavefront comparison of the graph's samples with a packed copy's state words (for calls without a prologue).
int launch_ids_sample_check(hipStream_t stream, const int32_t *col, int64_t n, const int32_t *pp, int64_t P,
                            const unsigned long long *state, int32_t *stale_flag, int32_t seq);
// Forgets the plans whose column_index starts at this address (all plans when null).  -> number of plans dropped.
// deferred: the device buffers are not freed now (no synchronisation: safe from a finalizer on any thread, during a
// stream capture) but at the next gnna_prepare_graph / gnna_release_graph / plan allocation with no launch in flight.
int release_slice_plans(const void *column_index, bool deferred);
void drain_dead_buffers();
// An aggregation call holds plan buffers between looking them up and enqueueing its kernels: dropped plans are not freed
// in between (RAII: LaunchGuard).
void begin_launch();
void end_launch();
struct LaunchGuard {
    LaunchGuard() { begin_launch(); }
    ~LaunchGuard() { end_launch(); }
    LaunchGuard(const LaunchGuard &) = delete;
    LaunchGuard &operator=(const LaunchGuard &) = delete;
};
// Number of phases of the sliced schedule from the statistics of the partition (gnna_agg.hip).
// Where to gather `dim`-float source rows from: the caller's layout or a staged copy in the stream's scratch (gnna_agg.hip).
int stage_rows_for_gather(DeviceState *ds, hipStream_t stream, const gnna_tuning &tune, const float *input, int64_t ld_in,
                          int64_t num_in_rows, int dim, int64_t est_edges, const float **X, int *ldx_out);
int choose_slices(const SlicePlanStats &st, size_t x_bytes, int S, uint32_t slice_rows, int64_t num_out_rows,
                  bool square, bool hinted_scattered);
// Events on the launch path that the contract promises not to happen after gnna_prepare_graph (gnna_runtime_counters).
enum { CTR_PLAN_BUILDS = 0, CTR_LAUNCH_SYNCS = 1, CTR_LAUNCH_FREES = 2, CTR_LAUNCH_MALLOCS = 3, CTR_BACKOFF_SKIPS = 4,
       CTR_SWEEP_LAUNCHES = 5, CTR_PACK_BUILDS = 6, CTR_PACKED_LAUNCHES = 7, CTR_FULL_HASHES = 8, CTR_COUNT = 9 };
void count_event(int which);

struct StreamLaunch {
    int mode;                 // MODE_SAG, MODE_GIN (also the pre-scaled GCN form: GIN + row_scale) or MODE_GCN (per-edge)
    const float *X; const int32_t *col; const int32_t *pp; const int32_t *p2n; float *Y;
    const uint8_t *cnt;       // slice counts or nullptr (single phase)
    const float *row_scale;
    const float *deg_row; const float *deg_col;   // MODE_GCN
    const float *A = nullptr;                      // MODE_SDDMM: destination-side features (Y = edge_out)
    const int32_t *flag; int32_t seq; int32_t trust;
    int64_t P;
    int D, ldx, G, U, S, B;
    int lda = 0;                                   // MODE_SDDMM: row stride of A in floats (0: D)
    int ldy = 0;                                   // row stride of Y in floats (0: D)
    int win_lo = 0, win_hi = 0;                    // windowed call: the fine slices (= source windows) it covers; win_hi == S: the rest
    bool relu = false;                             // epilogue: out = max(out, 0)
    int64_t num_out_rows = 0;                      // rows of Y (the epilogue's whole-output pass)
    bool wide, plain_ok, xcd_remap;
    float eps;
    bool det = false;                              // deterministic schedule (ordered phase launches, no atomics)
    float *det_part = nullptr; int32_t *det_stamp = nullptr;   // [num_chunks][2][D] partial rows / [num_chunks][2] stamps
    const int32_t *ids_packed = nullptr; const uint32_t *item_off = nullptr;   // packed ids of a prepared graph for (B, G)
    const int32_t *packed_stale = nullptr;   // *packed_stale == seq: column_index no longer matches the copy, read column_index
};
int launch_stream(const StreamLaunch &a, hipStream_t stream);
// The ReLU epilogue over the whole output on its own (a call that has nothing to aggregate but accumulates into `out`).
int launch_stream_epilogue_only(const StreamLaunch &a, hipStream_t stream);

// ---- destination-blocked sweep kernel (gnna_sweep.hip) ------------------------------------------------------
struct SweepLaunch {
    int mode;                 // MODE_SAG or MODE_GIN (also the pre-scaled GCN form: GIN + row_scale)
    const float *X; const int32_t *col; const int32_t *pp; const int32_t *p2n; float *Y;
    const uint8_t *cnt;       // slice counts (required)
    const float *row_scale;
    const int32_t *flag; int32_t seq; int32_t trust;
    uint32_t *sync;           // kXcds counters, 64 bytes apart, + the shared-row list's two header words: zero when the kernel starts
    int64_t P;
    int D, ldx, U, S, B;
    int ldy = 0;              // row stride of Y in floats (0: D)
    bool relu = false;        // epilogue: out = max(out, 0)
    int64_t num_out_rows = 0;
    int rounds = 0;           // sets per workgroup (0: from rows_with_edges and the accumulator capacity)
    int64_t rows_with_edges = 0;
    int slack = 0;            // soft-barrier slack in steps (0: built-in, >= 1000: none)
    int wgs_per_cu = 0;       // 2: two 16-wavefront workgroups per CU (64 VGPRs, half the accumulators each); else one
    bool dynamic = true;      // item pool with chunk locks (false: fixed share of the groups per wavefront)
    bool plain_ok;
    float eps;
    const int32_t *ids_packed = nullptr; const uint32_t *item_off = nullptr;   // packed ids for (B, 64 groups per chunk)
    const int32_t *packed_stale = nullptr;
};
bool sweep_supports(int mode, int dim, size_t x_bytes);
int sweep_acc_rows(int dim, int wgs_per_cu);   // destination rows a workgroup's LDS accumulators hold at this width
int launch_sweep(DeviceState *ds, const SweepLaunch &a, hipStream_t stream);
// Phases of the sweep kernel when the library picks it on its own (gnna_tuning.sweep = 0) for this call, else 0 (gnna_agg.hip).
int sweep_auto_phases(const gnna_tuning &t, int mode, int dim, size_t x_bytes, int64_t num_out_rows, int64_t num_in_rows,
                      const SlicePlanStats &st, int B, int num_cus, bool deterministic, int part_size);
// a call block of the sweep kernel: kXcds step counters 64 bytes apart, then the ReLU epilogue's
constexpr int kSweepListCap = 1023;    // list of row ranges the kernel did not store once: [count][overflow][(first, rows) ...]
constexpr int kSweepSlotWords = 8 * 16 + 2 + 2 * kSweepListCap;

// ---- optional per-call kernel timing (gnna_profile_begin/end) ---------------------------------------
// Returns the index of this call in the active profile (-1 when not profiling).
int profile_acquire_call(bool has_work);
// Records event `which` (0 before the prologue, 1 between, 2 after the aggregation) of call `call`.
void profile_record(int call, int which, hipStream_t stream);

}  // namespace gnna

#endif
