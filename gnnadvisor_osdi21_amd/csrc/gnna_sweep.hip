// gnna_sweep.hip -- the destination-blocked sweep form of the sliced neighbor-group aggregation.
// CDNA4 / gfx950 only.
//
// Same computation as stream_kernel (gnna_stream.hip; reference GNNAdvisor_kernel.cu:186-259, 620-689):
//   out[part2Node[g], :] += sum_{e in [partPtr[g], partPtr[g+1])} X[colidx[e], :]      (x eps / row factor)
// and the same slice plan (cum[f][g], gnna_stream.hip).  What is different is WHO keeps a destination row's
// partial sum between the source slices.
//
// Why: in the sliced schedule of stream_kernel a work item is (chunk, slice) and every (row, slice) piece ends in
// a flush of the row with memory-side float atomics on a zero-filled output.  More slices mean fewer L2 misses
// but more flushes -- on the Reddit-like headline (D = 64) 16 slices reach 83 % L2 hits and 4.9 GB of fabric
// traffic, yet lose to 8 slices (73 %, 7.9 GB) because of the atomics, and WRITE_SIZE is 8 x the output.
//
// Here ONE persistent 16-wavefront workgroup per CU owns a SET of consecutive neighbor-groups -- an equal share
// of the EDGES, found by a search in part_pointers -- for ALL source slices:
//
//   * the partial rows of the set (up to 512 rows of 64 floats) live in the CU's LDS (ds_add_f32; no global
//     atomics, nothing read back) while the workgroup walks the source slices 0 .. B-1;
//   * inside the workgroup the set's (64-group chunk, slice) items are drawn from a counter in LDS, slice-major
//     (or, `dynamic = 0`, every wavefront owns an equal contiguous share of the set's groups).  A wavefront holds
//     the chunk's bit of an LDS lock word while it runs an item, so the chunk's interior rows are touched by one
//     wavefront at a time (plain LDS read-add-write); only the rows that continue into a neighbouring chunk
//     (static: into a neighbouring wavefront's share -- from ANY chunk of the share) use ds_add_f32: an LDS
//     float atomic costs ~190 LDS cycles per wave-instruction -- with atomics for every row the LDS pipe was
//     busy 1.3 ms of a 2.0 ms launch;
//   * every row is written ONCE, after the last slice: a plain coalesced store when the set owns the row, one
//     atomic add when the row continues in a neighbouring set;
//   * the workgroups of an XCD (blockIdx % 8) walk the slices in step: a workgroup starts items of slice step t
//     only when every workgroup of its XCD has finished step t - slack (a per-XCD counter in memory, polled
//     with a bounded spin), so that the XCD's 4 MiB L2 holds about `slack` slices at a time and X is fetched
//     from the fabric (sets per workgroup) x 8 times per aggregation.  Only locality depends on that barrier,
//     never the result: a workgroup that waits too long stops waiting for the rest of the launch;
//   * rows beyond the accumulators' capacity (a set of many low-degree rows), and every row of a partition that
//     is not canonical, are flushed the old way (atomics per slice), which is correct for any partition (the sets of
//     such a partition are equal shares of the groups, not found by a search in its part_pointers).
//
// The per-(chunk, slice) item is the streaming kernel's: descriptors -> pieces -> load list -> U row loads
// always in flight -> fold at the last load of a row piece.
//
// Selected by the library (gnna_tuning.sweep = 0) where it measures faster than stream_kernel -- rows of 33..64 floats,
// long rows, a square Infinity-Cache-sized problem: sweep_auto_phases() in gnna_agg.hip, DESIGN.md 3 -- and by sweep = 1.
//
// (A first version of this file gave every WAVEFRONT its own set and 5.5 KiB of accumulators, 20 wavefronts per
// CU, barrier per wavefront: 2.25-2.5 ms on the Reddit-like headline without the barrier, 5-17 ms with it,
// against 1.59 ms for stream_kernel -- too few loads in flight, static sets, and a barrier whose unit is one
// latency chain.  DESIGN.md 3.)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>

#include "gnna.h"
#include "gnna_device.h"
#include "gnna_internal.h"

namespace gnna {
namespace {

constexpr int kSweepIdSlots = 512;   // column ids parked in LDS per wavefront and round of an item
constexpr int kSweepBlock = 1024;
constexpr int kSweepWaves = kSweepBlock / kWave;
constexpr int kSweepLockWords = 128;  // chunk locks of the dynamic pool: up to 4096 chunks per set

// LDS accumulator floats per workgroup: what 160 KiB leave next to the wavefronts' id lists.  WGS = workgroups per
// CU: 1 -> 16 wavefronts per CU with all of the LDS (128 / 112 KiB of accumulators, up to 128 VGPRs);
// 2 -> 32 wavefronts per CU (two workgroups of 60 / 44 KiB of accumulators each, 64 VGPRs): twice the latency chains
// in flight for the per-item round trips (descriptors -> ids -> rows), half the rows per set.
template <int LPR, int WGS> constexpr int acc_floats()
{
    return WGS == 1 ? (LPR >= 16 ? 32768 : 28672) : (LPR >= 16 ? 15360 : 11264);
}

// ds_add_f32 on a pointer that is known to point into LDS (an if-converted choice between an LDS and a global
// destination would otherwise become ONE flat atomic, which counts on vmcnt as well and stalls the load ring)
__device__ __forceinline__ void lds_add(float *p, float v)
{
    typedef __attribute__((address_space(3))) float lds_float;
    (void)__hip_atomic_fetch_add((lds_float *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

struct SweepParams {
    const float *X;
    const int32_t *col;
    const int32_t *pp;
    const int32_t *p2n;
    float *Y;
    const uint8_t *cnt;        // [S-1][P] cumulative slice counts
    const float *row_scale;    // MODE_GIN: optional per-destination-row factor on top of eps
    const int32_t *flag;       // *flag == seq  <=>  partition is NOT canonical
    uint32_t *sync;            // kXcds step counters of this call, 64 bytes apart, zero at launch
    int64_t P;
    int32_t seq;
    int32_t trust;
    int32_t D;
    int32_t ldx;
    int32_t ldy;               // row stride of Y in floats
    int32_t relu;              // 1: a row the set owns (written once, plain store) is written as max(x, 0)
    int32_t S;
    int32_t B;
    int32_t rounds;            // sets per workgroup
    int32_t plain_ok;          // 1: rows owned by one set are written with plain stores (out is not accumulated into)
    int32_t dynamic;           // 1: the wavefronts of a workgroup draw (chunk, slice) items from a pool (chunk locks in LDS)
                               // 0: every wavefront owns a fixed share of the set's groups
    int32_t slack;             // a workgroup starts step t once every workgroup of its XCD has finished step t - slack
                               // (1 = strict barrier); >= 1000: no synchronisation
    float eps;
    // packed column ids of a prepared graph for (B, 64 groups per chunk) -- see gnna_stream.hip -- or null.  With them the
    // sets start at multiples of 64 groups, so that a set's chunks are the global chunks the copy is laid out by.
    const int32_t *ids_packed;
    const uint32_t *item_off;
    const int32_t *packed_stale;   // *packed_stale == seq: column_index changed since the copy was made, read column_index
    int64_t num_chunks;        // ceil(P / 64)
};

// First index g in [0, P] with pp[g] >= target (pp non-decreasing, pp[P] >= target), searched 64 ways per round
// trip by one wavefront.
__device__ __forceinline__ int64_t lower_bound64(const int32_t *__restrict__ pp, int64_t P, int64_t target, int lane)
{
    int64_t lo = 0, hi = P;
    while (hi > lo) {
        const int64_t span = hi - lo;
        const int64_t idx = lo + (span * lane) / kWave;                  // lo <= idx < hi
        const bool below = (int64_t)pp[idx] < target;
        const int c = __popcll(__ballot(below));                        // a prefix of the lanes
        if (c == 0) { hi = lo; break; }
        const int64_t new_lo = lo + (span * (c - 1)) / kWave + 1;
        const int64_t new_hi = c < kWave ? lo + (span * c) / kWave : hi;
        lo = new_lo;
        hi = new_hi > new_lo ? new_hi : new_lo;
    }
    return lo;
}

// First group of set i of num_sets (one wavefront; every lane returns it): an equal share of the EDGES, found by a search
// in part_pointers -- for a partition that is not canonical (no search can be trusted on its part_pointers; every group
// then flushes per slice anyway) an equal share of the GROUPS.  With packed ids the sets start at multiples of 64 groups,
// so that a set's chunks are the global chunks the copy is laid out by.
__device__ __forceinline__ int64_t sweep_set_start(int64_t i, int64_t num_sets, int64_t nnz, const int32_t *__restrict__ pp,
                                                   int64_t P, bool canonical, bool align64, int lane)
{
    // (nnz < 2^31 and sets < 2^31: the products fit 64 bits; P < 2^31 groups)
    const int64_t target = i >= num_sets ? nnz : (nnz * i) / num_sets;
    int64_t g = i <= 0 ? 0 : (i >= num_sets ? P : (canonical ? lower_bound64(pp, P, target, lane) : (P * i) / num_sets));
    if (align64 && g < P) g &= ~(int64_t)(kWave - 1);
    return g;
}

// The ReLU epilogue's second half: rows the sweep kernel ADDED to the output instead of storing them once -- a row two sets
// share (float atomics from both), the rows of a set beyond its accumulators (flushed per slice) -- are clamped here,
// behind the kernel.  The kernel lists them itself while it runs, as (first row, rows) ranges of at most 16 rows:
// list[0] = number of ranges, list[2 + 2 i], list[3 + 2 i] = range i; list[1] is raised when the list is full.  Then, and
// for a partition that is not canonical or a call that accumulates into an existing output (`whole`), the whole output is
// clamped.  (Measured on the headline, D = 64: aggregation 1.431 ms, with this epilogue 1.442, aggregation + torch.relu 1.451.)
__global__ void __launch_bounds__(kBlock)
sweep_relu_fixup_kernel(float *__restrict__ Y, int64_t N, int D, int ldy, const uint32_t *__restrict__ list,
                        const int32_t *flag, int32_t seq, int32_t trust, int whole)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const bool canonical = trust || (*flag != seq);
    auto clamp_row = [&](int64_t r) {
        float *row = Y + (size_t)r * (size_t)ldy;
        for (int i = lane; i < D; i += kWave) row[i] = fmaxf(row[i], 0.f);
    };
    if (whole || !canonical || list[1] != 0u) {
        for (int64_t r = wave; r < N; r += nwaves) clamp_row(r);
        return;
    }
    const uint32_t n = list[0] < (uint32_t)kSweepListCap ? list[0] : (uint32_t)kSweepListCap;
    for (int64_t i = wave; i < (int64_t)n; i += nwaves) {    // one wavefront per range
        const int64_t first = (int64_t)list[2 + 2 * i], cnt = (int64_t)list[3 + 2 * i];
        for (int64_t r = first; r < first + cnt && r < N; r++) clamp_row(r);
    }
}

template <int LPR, int MODE, int U, int WGS>
__global__ void __launch_bounds__(kSweepBlock) __attribute__((amdgpu_waves_per_eu(4 * WGS, 4 * WGS)))
sweep_kernel(const SweepParams p)
{
    typedef typename VecOf<4>::T VT;
    typedef typename VecOf<4>::M MT;
    constexpr int RPI = kWave / LPR;                                   // neighbor rows per wave-wide load
    constexpr int RL = (kSweepIdSlots / RPI < kWave) ? kSweepIdSlots / RPI : kWave;   // loads per round
    static_assert(RL % U == 0, "a round is a whole number of batches");
    constexpr int ACC = acc_floats<LPR, WGS>();
    __shared__ float s_acc[ACC];                          // partial rows of the workgroup's set
    __shared__ uint32_t s_off[kSweepWaves][RL * RPI];     // per wavefront: the round's list slots as byte offsets into X
    __shared__ int64_t s_g[2];                            // the set's group range
    __shared__ int s_ctl[8];                              // 0 next item, 1 items / wavefront-steps done, 2 last seen XCD counter, 3 gave up
    __shared__ unsigned s_lock[kSweepLockWords];          // dynamic pool: one bit per chunk of the set (a chunk's rows are one wavefront's at a time)

    const int lane = threadIdx.x & (kWave - 1);
    const int wib = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int lslot = lane / LPR;
    const int c = lane % LPR;
    const int D = p.D;
    const bool canonical = p.trust || (*p.flag != p.seq);
    const bool stale_ids = p.ids_packed != nullptr && *p.packed_stale == p.seq;
    const char *xbase = reinterpret_cast<const char *>(p.X);
    const uint32_t row_bytes32 = (uint32_t)p.ldx * 4u;
    uint32_t *offs = s_off[wib];
    const int CAP = ACC / D;                                            // accumulator rows (>= 1: the launcher checks D)
    const unsigned long long upto = (2ull << lane) - 1ull;             // lanes <= lane
    const unsigned long long above = ~upto;                             // lanes > lane

    // lane c owns the 4 floats starting at dcol; a ragged last piece is shifted back to end at D and its first
    // `shift` components repeat the predecessor's (they are loaded but never added)
    const int piece = c * 4;
    const bool cvalid = piece < D;
    int dcol = piece, shift = 0;
    if (piece + 4 > D && cvalid) { dcol = D - 4; shift = piece - dcol; }
    const uint32_t col_off = (uint32_t)(cvalid ? dcol : 0) * 4u;
    // after fold_row a lane of a row of <= 64 floats holds component lane >> 4 of its piece
    const int comp = lane >> 4;
    const bool add_lane = LPR <= 16 ? ((lane & 15) < LPR && cvalid && comp >= shift) : (lslot == 0 && cvalid);

    const int xcd = (int)(blockIdx.x & (kXcds - 1));
    const int nbx = (int)(gridDim.x >> 3);                             // workgroups per XCD
    const int bx = (int)(blockIdx.x >> 3);
    uint32_t *ctr = p.sync + xcd * 16;
    const bool syncing = p.slack < 1000 && nbx > 1;
    const int B = p.B, R = p.rounds;
    const int64_t nnz = (int64_t)p.pp[p.P];
    const int64_t num_sets = (int64_t)gridDim.x * R;

    if (threadIdx.x == 0) { s_ctl[2] = 0; s_ctl[3] = 0; }

    for (int r = 0; r < R; r++) {
        // ---- the set: an equal share of the edges, XCD-major, then workgroup, then round -------------------
        const int64_t set = ((int64_t)xcd * nbx + bx) * R + r;
        if (wib < 2) {
            const int64_t g = sweep_set_start(set + wib, num_sets, nnz, p.pp, p.P, canonical, p.ids_packed != nullptr, lane);
            if (lane == 0) s_g[wib] = g;
        }
        if (threadIdx.x == 0) { s_ctl[0] = 0; s_ctl[1] = 0; }
        __syncthreads();
        const int64_t g_lo = s_g[0], g_hi = s_g[1];
        int row_first = 0, row_last = -1, set_prev_row = -1, set_next_row = -1;
        if (g_hi > g_lo) {
            row_first = p.p2n[g_lo];
            row_last = p.p2n[g_hi - 1];
            if (g_lo > 0) set_prev_row = p.p2n[g_lo - 1];
            if (g_hi < p.P) set_next_row = p.p2n[g_hi];
        }
        int nrows = 0;                                                   // rows of the set kept in LDS
        if (canonical && row_last >= row_first) nrows = row_last - row_first + 1 < CAP ? row_last - row_first + 1 : CAP;
        if (p.relu && canonical && threadIdx.x == 0 && g_hi > g_lo) {
            // ReLU epilogue: the rows this set does not finish alone go on the call's list for the pass behind the kernel
            uint32_t *list = p.sync + kXcds * 16;
            auto push = [&](int first, int cnt) {
                const uint32_t idx = atomicAdd(&list[0], 1u);
                if (idx < (uint32_t)kSweepListCap) { list[2 + 2 * idx] = (uint32_t)first; list[3 + 2 * idx] = (uint32_t)cnt; }
                else list[1] = 1u;
            };
            if (set_prev_row == row_first) push(row_first, 1);
            const int over_rows = row_last - row_first + 1 - CAP;       // rows beyond the accumulators, in ranges of <= 16
            if (over_rows > 2048) list[1] = 1u;
            else
                for (int r0 = 0; r0 < over_rows; r0 += 16) push(row_first + CAP + r0, over_rows - r0 < 16 ? over_rows - r0 : 16);
        }
        for (int i = threadIdx.x; i < nrows * D; i += kSweepBlock) s_acc[i] = 0.f;
        __syncthreads();

        // Two ways to hand the set's (chunk, slice) items to the 16 wavefronts:
        //  * static (p.dynamic == 0): every wavefront owns an equal, contiguous share of the set's groups for all slices;
        //    a destination row is then one wavefront's, except the rows at the borders of its share;
        //  * dynamic: the items are a pool in slice order that the wavefronts drain (an LDS counter) -- a wavefront that
        //    drew a short item draws the next.  A bit per chunk in LDS makes a chunk's rows one wavefront's at a time
        //    (items of one chunk in consecutive slices are ~nchunks draws apart, so the lock is almost never contended);
        //    only the rows a chunk shares with its neighbours are added atomically.
        const int64_t ng_set = g_hi - g_lo;
        const int nchunks = (int)((ng_set + kWave - 1) / kWave);
        const bool dyn = p.dynamic != 0 && nchunks <= kSweepLockWords * 32;
        const int64_t gw_lo = dyn ? g_lo : g_lo + (ng_set * wib) / kSweepWaves;
        const int64_t gw_hi = dyn ? g_hi : g_lo + (ng_set * (wib + 1)) / kSweepWaves;
        const int nch_w = (int)((gw_hi - gw_lo + kWave - 1) / kWave);        // chunks this wavefront may work on
        const int per_step = nch_w > 0 ? nch_w : 1;
        if (dyn)
            for (int i = threadIdx.x; i < kSweepLockWords; i += kSweepBlock) s_lock[i] = 0u;
        __syncthreads();
        const int total_items = dyn ? nchunks * B : per_step * B;
        int it = 0, last_T = -1;
        while (true) {
            int item = it++;
            if (dyn) {
                if (lane == 0) item = __hip_atomic_fetch_add(&s_ctl[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                item = __builtin_amdgcn_readfirstlane(item);
            }
            if (item >= total_items) break;
            const int t = item / per_step;                              // items come in slice order
            const int chunk = item - t * per_step;
            const int T = r * B + t;
            const int64_t g0 = gw_lo + (int64_t)chunk * kWave;
            const int ng = nch_w > 0 ? (int)(gw_hi - g0 < (int64_t)kWave ? gw_hi - g0 : (int64_t)kWave) : 0;

            // ---- soft barrier: every workgroup of the XCD has finished step T - slack -------------------------
            if (syncing && T >= p.slack && T != last_T) {
                const uint32_t need = (uint32_t)nbx * (uint32_t)(T - p.slack + 1);
                const int seen = __hip_atomic_load(&s_ctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const int gave_up = __hip_atomic_load(&s_ctl[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if ((int32_t)((uint32_t)__builtin_amdgcn_readfirstlane(seen) - need) < 0 && !__builtin_amdgcn_readfirstlane(gave_up)) {
                    const unsigned long long t0 = wall_clock64();
                    while (true) {
                        const uint32_t now = __builtin_amdgcn_readfirstlane(
                            __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                        if ((int32_t)(now - need) >= 0) {
                            if (lane == 0) __hip_atomic_store(&s_ctl[2], (int)now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            break;
                        }
                        if (wall_clock64() - t0 > 20000ull) {           // 200 us: this workgroup stops waiting for good
                            if (lane == 0) __hip_atomic_store(&s_ctl[3], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(4);
                    }
                }
            }
            last_T = T;
            // rows that continue outside this item's groups: the only accumulator rows another wavefront may add to
            int wave_prev_row = -1, wave_next_row = -1;
            if (ng > 0) {
                // static share: the rows at the SHARE's borders, in every chunk of it (a long row can reach from an
                // interior chunk to the share's end); dynamic pool: the rows at the chunk's own borders
                const int64_t lo = dyn ? g0 : gw_lo, hi = dyn ? g0 + ng : gw_hi;
                if (lo > g_lo) wave_prev_row = p.p2n[lo - 1];
                if (hi < g_hi) wave_next_row = p.p2n[hi];
            }
            if (dyn && lane == 0) {   // the chunk's rows are this wavefront's until the item is done
                const unsigned bit = 1u << (chunk & 31);
                while (__hip_atomic_fetch_or(&s_lock[chunk >> 5], bit, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) & bit)
                    __builtin_amdgcn_s_sleep(2);
            }

            if (ng > 0)
            {
                const int f_lo = t * p.S / B, f_hi = (t + 1) * p.S / B;

                // ---- 1. descriptors (all loads in flight together) -------------------------------------
                const bool gl = lane < ng;
                const int my_row = gl ? p.p2n[g0 + lane] : -1;
                const int pa = gl ? p.pp[g0 + lane] : 0;
                const int pb = gl ? p.pp[g0 + lane + 1] : 0;
                int cum_lo = 0, cum_hi = 0x7fffffff;
                if (gl) {
                    if (f_lo > 0) cum_lo = p.cnt[(size_t)(f_lo - 1) * (size_t)p.P + (size_t)(g0 + lane)];
                    if (f_hi < p.S) cum_hi = p.cnt[(size_t)(f_hi - 1) * (size_t)p.P + (size_t)(g0 + lane)];
                }
                const bool packed = dyn && p.ids_packed != nullptr && !stale_ids;
                const uint32_t item_base = packed ? p.item_off[(size_t)t * (size_t)p.num_chunks + (size_t)(g0 >> 6)] : 0u;
                const int32_t *__restrict__ ids = packed ? p.ids_packed : p.col;
                const int len = pb > pa ? pb - pa : 0;
                const int beg = cum_lo < len ? cum_lo : len;
                int end = cum_hi < len ? cum_hi : len;
                end = end > beg ? end : beg;
                const int n_own = gl ? end - beg : 0;  // edges of this group in this phase
                if (__ballot(n_own > 0) != 0) {
                    // destination-row segments; pieces of consecutive groups of one row that are adjacent in the
                    // edge array are merged into one piece, headed by the first
                    const int up_row = __shfl_up(my_row, 1);
                    const bool seg_start = gl && (lane == 0 || my_row != up_row || !canonical);
                    const unsigned long long SS = __ballot(seg_start);
                    const int up_end = __shfl_up(pa + end, 1), up_n = __shfl_up(n_own, 1);
                    const bool cont = n_own > 0 && !seg_start && up_n > 0 && (packed || up_end == pa + beg);
                    const unsigned long long NE = __ballot(n_own > 0 && !cont);        // piece heads
                    const int n_cum = wave_inclusive_scan(n_own);
                    const unsigned long long heads_above = NE & above;
                    const int next_head = heads_above ? __builtin_ctzll(heads_above) : 64;
                    const int chain_cum = __shfl(n_cum, next_head - 1);
                    const int n = (n_own > 0 && !cont) ? chain_cum - n_cum + n_own : 0;
                    const unsigned long long between = SS & above & (next_head < 63 ? ((2ull << next_head) - 1ull) : ~0ull);
                    const bool last_in_seg = n > 0 && (next_head == 64 || between != 0);
                    const int aslot = my_row - row_first;
                    const int over = (aslot < 0 || aslot >= nrows) ? 1 : 0;   // not in the accumulators: flushed per slice
                    const int edge = (my_row == wave_prev_row || my_row == wave_next_row) ? 4 : 0;   // shared with a neighbour wavefront

                    // non-empty pieces compacted to lanes 0 .. Rn-1
                    const int rank = __popcll(NE & (upto >> 1));
                    const int Rn = __popcll(NE);
                    const int dstl = (n > 0 ? rank : 63) << 2;
                    const int c_pbeg = __builtin_amdgcn_ds_permute(dstl, packed ? (int)item_base + (n_cum - n_own) : pa + beg);
                    const int t_n = __builtin_amdgcn_ds_permute(dstl, n);
                    const int c_n = lane < Rn ? t_n : 0;
                    const int c_meta = __builtin_amdgcn_ds_permute(dstl, ((over ? my_row : aslot) << 3) | edge | (last_in_seg ? 2 : 0) | over);
                    const int c_nl = (c_n + RPI - 1) / RPI;
                    const int c_offI = wave_inclusive_scan(c_nl);
                    const int c_offX = c_offI - c_nl;
                    const int L = __builtin_amdgcn_readlane(c_offI, kWave - 1);

                    VT acc = vzero<4>();
                    // ---- 2. a round's loads: lane j describes load r0 + j and fetches its RPI column ids.  The description
                    // of round r + 1 -- ids included -- is made BEFORE round r is streamed: its id loads travel behind the
                    // row loads of round r instead of standing alone between two rounds (a work item of the sliced schedule
                    // is 2-4 rounds; with 16 wavefronts per CU nothing else covers that round trip).
                    struct Round {
                        uint32_t o[RPI];              // the load's row offsets (bytes)
                        int v_j, k_meta;
                        unsigned long long FL, TM;
                        int nr;
                    };
                    auto describe = [&](int r0) -> Round {
                        Round rd;
                        const unsigned long long below = __ballot(lane < Rn && c_offI <= r0);
                        unsigned long long inwin = __ballot(lane < Rn && c_offI > r0 && c_offI <= r0 + RL - 1);
                        unsigned long long E = 0;
                        while (inwin) {
                            const int kk = __builtin_ctzll(inwin);
                            inwin &= inwin - 1;
                            E |= 1ull << (__builtin_amdgcn_readlane(c_offI, kk) - r0);
                        }
                        const int kp = __popcll(below) + __popcll(E & upto);
                        const int k_offX = __shfl(c_offX, kp), k_pbeg = __shfl(c_pbeg, kp), k_n = __shfl(c_n, kp);
                        rd.k_meta = __shfl(c_meta, kp);
                        const int J = r0 + lane;
                        const bool active = J < L && lane < RL;
                        const int i = J - k_offX;
                        const int e_j = k_pbeg + i * RPI;
                        int v_j = active ? k_n - i * RPI : 0;
                        const bool fl_j = active && (rd.k_meta & 2) && v_j <= RPI;   // last load of the last piece of its row
                        v_j = v_j > RPI ? RPI : v_j;
                        rd.v_j = v_j;
                        rd.FL = __ballot(fl_j);
                        rd.TM = __ballot(active && v_j < RPI);   // loads with padded slots
                        rd.nr = (L - r0) < RL ? (L - r0) : RL;
#pragma unroll
                        for (int q = 0; q < RPI; q++) rd.o[q] = 0u;
                        if (lane < RL) {
                            if (v_j == RPI) {
                                if constexpr (RPI >= 4) {
                                    typedef int i32x4 __attribute__((ext_vector_type(4)));
                                    typedef i32x4 i32x4u __attribute__((aligned(4)));
#pragma unroll
                                    for (int s4 = 0; s4 < RPI; s4 += 4) {
                                        const i32x4 tt = *reinterpret_cast<const i32x4u *>(ids + e_j + s4);
                                        rd.o[s4] = (uint32_t)tt[0]; rd.o[s4 + 1] = (uint32_t)tt[1]; rd.o[s4 + 2] = (uint32_t)tt[2]; rd.o[s4 + 3] = (uint32_t)tt[3];
                                    }
                                } else {
#pragma unroll
                                    for (int q = 0; q < RPI; q++) rd.o[q] = (uint32_t)ids[e_j + q];
                                }
                            } else {
                                const uint32_t first = v_j > 0 ? (uint32_t)ids[e_j] : 0u;
#pragma unroll
                                for (int q = 0; q < RPI; q++) {
                                    rd.o[q] = first;
                                    if (q > 0 && q < v_j) rd.o[q] = (uint32_t)ids[e_j + q];
                                }
                            }
                        }
                        return rd;
                    };
                    Round cur = describe(0);
                    for (int r0 = 0; r0 < L; r0 += RL) {
                        if (lane < RL) {
#pragma unroll
                            for (int q = 0; q < RPI; q++) offs[lane * RPI + q] = cur.o[q] * row_bytes32;
                        }
                        const bool more = r0 + RL < L;
                        Round nxt = cur;
                        if (more) nxt = describe(r0 + RL);
                        const int v_j = cur.v_j, k_meta = cur.k_meta, nr = cur.nr;
                        const unsigned long long FL = cur.FL, TM = cur.TM;

                        // ---- 3. stream: U unpredicated row loads always in flight -----------------------------
                        auto row_ptr = [&](uint32_t o) -> const MT * {
                            return reinterpret_cast<const MT *>(xbase + (o + col_off));
                        };
                        const int nb = (nr + U - 1) / U;
                        VT v[U];
#pragma unroll
                        for (int u = 0; u < U; u++) v[u] = *row_ptr(offs[u * RPI + lslot]);
                        auto consume = [&](int u, int j) {
                            if ((TM >> j) & 1ull) {
                                const int vj = __builtin_amdgcn_readlane(v_j, j);
                                if (lslot >= vj) v[u] = vzero<4>();
                            }
                            acc += v[u];
                            if ((FL >> j) & 1ull) {
                                const int meta = __builtin_amdgcn_readlane(k_meta, j);
                                const VT rr = fold_row<LPR, MODE_SAG>(acc, 1.f);
                                if (!(meta & 1)) {
                                    // the set's own accumulator row: LDS, no memory traffic.  A row inside this wavefront's
                                    // share is its own (plain read-add-write); a border row is shared with one neighbour
                                    float *dst = s_acc + (meta >> 3) * D + dcol;
                                    if (meta & 4) {
                                        if constexpr (LPR <= 16) {
                                            if (add_lane) lds_add(dst + comp, rr[0]);
                                        } else {
                                            if (add_lane) {
#pragma unroll
                                                for (int q = 0; q < 4; q++)
                                                    if (q >= shift) lds_add(dst + q, rr[q]);
                                            }
                                        }
                                    } else {
                                        if constexpr (LPR <= 16) {
                                            if (add_lane) dst[comp] += rr[0];
                                        } else {
                                            if (add_lane) {
#pragma unroll
                                                for (int q = 0; q < 4; q++)
                                                    if (q >= shift) dst[q] += rr[q];
                                            }
                                        }
                                    }
                                } else {
                                    // not in the accumulators: add to the (zero-filled) output now, as stream_kernel does
                                    const int64_t row = meta >> 3;
                                    float scale = 1.f;
                                    if constexpr (MODE == MODE_GIN) {
                                        scale = p.eps;
                                        if (p.row_scale) scale *= p.row_scale[row];
                                    }
                                    float *dst = p.Y + (size_t)row * (size_t)p.ldy + dcol;
                                    if constexpr (LPR <= 16) {
                                        if (add_lane) unsafeAtomicAdd(dst + comp, rr[0] * scale);
                                    } else {
                                        if (add_lane) {
#pragma unroll
                                            for (int q = 0; q < 4; q++)
                                                if (q >= shift) unsafeAtomicAdd(dst + q, rr[q] * scale);
                                        }
                                    }
                                }
                                acc = vzero<4>();
                            }
                        };
#pragma unroll 1
                        for (int b = 0; b + 1 < nb; b++) {
                            const int jn = (b + 1) * U;
                            uint32_t nn[U];
#pragma unroll
                            for (int u = 0; u < U; u++) nn[u] = offs[(jn + u) * RPI + lslot];
#pragma unroll
                            for (int u = 0; u < U; u++) {
                                consume(u, b * U + u);
                                v[u] = *row_ptr(nn[u]);
                            }
                        }
#pragma unroll
                        for (int u = 0; u < U; u++) {      // the last batch only consumes
                            const int j = (nb - 1) * U + u;
                            if (j < nr) consume(u, j);
                        }
                        cur = nxt;
                    }
                }
            }
            if (dyn && lane == 0)
                (void)__hip_atomic_fetch_and(&s_lock[chunk >> 5], ~(1u << (chunk & 31)), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            // ---- arrival: the workgroup has finished a slice step when all its items (dynamic) / all its wavefronts'
            // shares (static) of that step are done
            if (syncing && lane == 0 && (dyn || chunk == per_step - 1)) {
                const int done = __hip_atomic_fetch_add(&s_ctl[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 1;
                if (done % (dyn ? nchunks : kSweepWaves) == 0)
                    (void)__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (dyn && nchunks == 0 && syncing && threadIdx.x == 0)        // nothing to do: still counts as arrived
            (void)__hip_atomic_fetch_add(ctr, (uint32_t)B, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();

        // ---- every accumulator row is written once ----------------------------------------------------------
        for (int q = wib; q < nrows; q += kSweepWaves) {
            const int row = row_first + q;
            const bool shared = row == set_prev_row || row == set_next_row;
            const bool use_atomic = shared || !p.plain_ok;
            float scale = 1.f;
            if constexpr (MODE == MODE_GIN) {
                scale = p.eps;
                if (p.row_scale) scale *= p.row_scale[row];
            }
            float *dst = p.Y + (size_t)row * (size_t)p.ldy;
            const float *src = s_acc + q * D;
            for (int i = lane; i < D; i += kWave) {
                const float val = src[i] * scale;
                // (fused ReLU: where the row is complete at this, its only, write; shared rows: sweep_relu_fixup_kernel)
                if (!use_atomic) __builtin_nontemporal_store(p.relu ? fmaxf(val, 0.f) : val, dst + i);
                else unsafeAtomicAdd(dst + i, val);
            }
        }
        __syncthreads();
    }
}

typedef void (*SweepKernel)(const SweepParams);

template <int LPR, int MODE, int WGS>
SweepKernel pick_sweep_u(int u)
{
    constexpr int RPI = kWave / LPR;
    constexpr int RL = (kSweepIdSlots / RPI < kWave) ? kSweepIdSlots / RPI : kWave;
    if constexpr (RL % 16 == 0 && WGS == 1) {
        if (u >= 16) return sweep_kernel<LPR, MODE, 16, WGS>;
    }
    if constexpr (RL % 8 == 0 && WGS == 1) {
        if (u >= 8) return sweep_kernel<LPR, MODE, 8, WGS>;
    }
    return sweep_kernel<LPR, MODE, 4, WGS>;
}

template <int MODE, int WGS>
SweepKernel pick_sweep(int lpr, int u)
{
    switch (lpr) {
    case 4: return pick_sweep_u<4, MODE, WGS>(u);
    case 8: return pick_sweep_u<8, MODE, WGS>(u);
    case 16: return pick_sweep_u<16, MODE, WGS>(u);
    default: return pick_sweep_u<32, MODE, WGS>(u);
    }
}

}  // namespace

int sweep_acc_rows(int dim, int wgs)
{
    const int pieces = (dim + 3) / 4;
    const int acc = wgs == 1 ? (pieces <= 8 ? acc_floats<8, 1>() : acc_floats<16, 1>())
                             : (pieces <= 8 ? acc_floats<8, 2>() : acc_floats<16, 2>());
    return std::max(1, acc / std::max(1, dim));
}

bool sweep_supports(int mode, int dim, size_t x_bytes)
{
    return (mode == MODE_SAG || mode == MODE_GIN) && dim >= 4 && dim <= 128 && x_bytes <= 0xffffffffull;
}

int launch_sweep(DeviceState *ds, const SweepLaunch &a, hipStream_t stream)
{
    SweepParams p;
    p.X = a.X; p.col = a.col; p.pp = a.pp; p.p2n = a.p2n; p.Y = a.Y; p.cnt = a.cnt; p.row_scale = a.row_scale;
    p.flag = a.flag; p.seq = a.seq; p.trust = a.trust; p.sync = a.sync;
    p.P = a.P; p.D = a.D; p.ldx = a.ldx; p.S = a.S; p.B = a.B; p.plain_ok = a.plain_ok ? 1 : 0; p.eps = a.eps;
    p.ldy = a.ldy > 0 ? a.ldy : a.D;
    p.relu = (a.relu && a.plain_ok) ? 1 : 0;
    p.slack = a.slack > 0 ? a.slack : 2;
    p.ids_packed = a.packed_stale ? a.ids_packed : nullptr; p.item_off = a.item_off; p.packed_stale = a.packed_stale;
    p.num_chunks = (a.P + kWave - 1) / kWave;
    p.dynamic = a.dynamic ? 1 : 0;
    int lpr = 4;
    const int pieces = (a.D + 3) / 4;
    while (lpr < 32 && lpr < pieces) lpr <<= 1;
    const int wgs = a.wgs_per_cu == 2 ? 2 : 1;
    SweepKernel k = wgs == 1 ? (a.mode == MODE_GIN ? pick_sweep<MODE_GIN, 1>(lpr, a.U) : pick_sweep<MODE_SAG, 1>(lpr, a.U))
                             : (a.mode == MODE_GIN ? pick_sweep<MODE_GIN, 2>(lpr, a.U) : pick_sweep<MODE_SAG, 2>(lpr, a.U));
    // persistent grid: `wgs` workgroups per CU (together all of the CU's LDS), the same number on every XCD
    const int cus_per_xcd = std::max(1, ds->num_cus / kXcds);
    const unsigned grid = (unsigned)(cus_per_xcd * kXcds * wgs);
    // sets per workgroup: as few as keep a set's rows (on average, with some head room) inside the accumulators
    const int cap = sweep_acc_rows(a.D, wgs);
    int R = a.rounds;
    if (R <= 0) {
        const double rows = (double)std::max<int64_t>(1, a.rows_with_edges);
        R = (int)std::max(1.0, std::ceil(rows / ((double)grid * 0.9 * (double)cap)));
    }
    p.rounds = std::min(R, 4096);
    hipLaunchKernelGGL(k, dim3(grid), dim3(kSweepBlock), 0, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GNNA_ERR_HIP, "sweep launch: %s", hipGetErrorString(e));
    count_event(CTR_SWEEP_LAUNCHES);
    if (a.relu) {
        const bool whole = !a.plain_ok;
        // (which of the two jobs it is, is decided on the device -- the list may have overflowed: always a grid that can
        // do the whole output; idle blocks cost nothing next to the aggregation)
        const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((a.num_out_rows + kWavesPerBlock - 1) / kWavesPerBlock, 256 * 8));
        hipLaunchKernelGGL(sweep_relu_fixup_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, stream, a.Y, a.num_out_rows, a.D, p.ldy,
                           a.sync + kXcds * 16, a.flag, a.seq, a.trust, whole ? 1 : 0);
        e = hipGetLastError();
        if (e != hipSuccess) return fail(GNNA_ERR_HIP, "epilogue launch: %s", hipGetErrorString(e));
    }
    return GNNA_OK;
}

}  // namespace gnna
