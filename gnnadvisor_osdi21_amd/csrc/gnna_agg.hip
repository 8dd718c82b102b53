// gnna_agg.hip -- CDNA4 (gfx950) neighbor-group aggregation kernels + C-ABI launchers.
//
// Replaces the five CUDA kernels of the reference (GNNAdvisor/GNNConv/GNNAdvisor_kernel.cu:
// SAG :186-259, GCN fwd :324-415, GCN bwd :478-552, GIN fwd :620-689, GIN bwd :749-814) with
// one templated HIP kernel.  Not a translation: see DESIGN.md "Kernel".
//
// Shape of the computation (all three modes):
//   out[part2Node[p], :] += sum_{e in [partPtr[p], partPtr[p+1])} coef * X[colidx[e], :]
//
// CDNA4 mapping
//   * a 64-lane wavefront owns a *chunk* of G consecutive neighbor-groups.  Groups of one
//     destination row are adjacent (build_part emits them consecutively), so a run of
//     same-row groups is one contiguous edge segment that the wave reduces in registers;
//     only a row that continues into a neighbouring chunk needs atomics.
//   * lanes are laid out (slot, c): c = lane % LPR addresses a VEC-float piece of the
//     feature row, slot = lane / LPR addresses one of RPI = 64/LPR neighbor rows, so one
//     wave-wide global_load_dwordx4 fetches RPI complete, fully coalesced rows (D=64:
//     4 rows = 1 KiB per instruction).  U such loads are issued back-to-back before the
//     first add (memory-level parallelism; the kernel is gather-bandwidth bound).
//   * column ids are fetched 64 at a time with one coalesced non-temporal load and
//     handed to the slots with ds_bpermute; partial rows live in VGPRs (the reference's
//     shared-memory read-modify-write chain, .cu:245-249, is what it is bound by).
//   * slots are folded once per destination row as a reduce-scatter with v_permlane32_swap /
//     v_permlane16_swap (3 swaps + 3 adds leave every lane with one float of the row) and DPP
//     row rotations for narrow rows.
//   * flush: plain (non-temporal) vector store when the row is wholly owned by the wave,
//     hardware global_atomic_add_f32 only for rows shared with a neighbouring chunk.
//   * a prologue kernel zero-fills `out` and checks that the partition is canonical
//     (part2Node and partPtr non-decreasing); if it is not, every group is flushed with
//     atomics, which is correct for any partition.
//   * column phases (PHASED): the kernel is launched once per source-id range so that the slice of
//     X being gathered is cache resident; every run keeps a cursor (next unconsumed edge) between
//     the launches.  The windowed entry point runs a sub-range of those launches per call
//     (pipelined multi-GPU exchange).
//   * row staging: for widths whose rows straddle 128-byte lines (41, 47, 56 ...) and for the
//     pre-scaled GCN form the source rows are first copied into scratch with a line-friendly
//     row stride (and multiplied by their degree norm); the kernel takes the stride as `ldx`.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>

#include "gnna.h"
#include "gnna_device.h"
#include "gnna_internal.h"

namespace gnna {
namespace {

struct AggParams {
    const float *X;
    const int32_t *col;
    const float *deg_row;  // per destination row (GCN)
    const float *deg_col;  // per source row (GCN)
    const int32_t *pp;
    const int32_t *p2n;
    float *Y;
    int64_t P;           // number of neighbor-groups
    int64_t num_chunks;  // ceil(P / G)
    int64_t num_items;   // ceil(num_chunks / waves per block)
    int64_t items_per_xcd;
    const int32_t *flag; // *flag == seq  <=>  partition is NOT canonical
    int32_t seq;
    int32_t trust;
    int32_t D;
    int32_t ldx;  // row stride of X in floats (== D unless the rows were staged into a padded layout)
    int32_t G;
    int32_t xcd_remap;
    float eps;
    // column-phased schedule (num_phases > 1): this launch consumes, for every run, the
    // edges from the run's cursor up to the first source id >= phase_hi
    int32_t *cursor;   // [P] next unconsumed edge of the run starting at that group
    int32_t phase;
    int32_t num_phases;
    int32_t phase_hi;
    int32_t acc_in;  // 1: add to the existing contents of Y instead of overwriting (no zero-fill)
    const float *row_scale;  // MODE_GIN only: optional per-destination-row factor on top of eps
};

// ---- prologue: zero-fill + partition validation ------------------------------------------

__global__ void __launch_bounds__(kBlock)
prologue_kernel(float *__restrict__ Y, size_t n_floats, const int32_t *__restrict__ p2n,
                const int32_t *__restrict__ pp, int64_t P, int32_t *flag, int32_t seq, int validate, int zero_fill,
                const int32_t *__restrict__ chk_ids, int64_t chk_n, const unsigned long long *__restrict__ chk_sum,
                int32_t *stale_flag)
{
    // packed ids of a prepared graph: does column_index still look like the array the copy was made from?  (1024 samples:
    // catches a buffer that was rewritten, which is what happens when a promise of immutability is broken by accident.)
    // The launcher adds one block for it, so that the zero-fill does not wait behind the sampled loads.
    const unsigned fill_blocks = chk_ids ? gridDim.x - 1 : gridDim.x;
    if (blockIdx.x >= fill_blocks) {
        if (threadIdx.x < kWave) {
            const unsigned long long then = *chk_sum;
            const unsigned long long now = graph_checksum(chk_ids, chk_n, pp, P, (int)threadIdx.x);
            if (threadIdx.x == 0 && now != then) *stale_flag = seq;
        }
        return;
    }
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nthreads = (size_t)fill_blocks * blockDim.x;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    if (!zero_fill) {
        // accumulate mode: Y keeps its contents
    } else if ((reinterpret_cast<uintptr_t>(Y) & 15) == 0) {
        const size_t n4 = n_floats >> 2;
        f32x4 *Y4 = reinterpret_cast<f32x4 *>(Y);
        const f32x4 z = (f32x4)(0.f);
        for (size_t i = tid; i < n4; i += nthreads) Y4[i] = z;
        for (size_t i = (n4 << 2) + tid; i < n_floats; i += nthreads) Y[i] = 0.f;
    } else {
        for (size_t i = tid; i < n_floats; i += nthreads) Y[i] = 0.f;
    }
    if (validate) {
        bool bad = false;
        for (int64_t g = (int64_t)tid; g < P; g += (int64_t)nthreads) {
            if (pp[g + 1] < pp[g]) bad = true;
            if (g + 1 < P && p2n[g + 1] < p2n[g]) bad = true;
        }
        if (bad) *flag = seq;  // every writer stores the same value
    }
}

// ---- sparse prologue: zero only the rows nobody overwrites --------------------------------------
// A single pass of the streaming kernel over a canonical partition STORES every destination row that
// has an edge and is not shared between two work items; the zero-fill of such a row is written once and
// overwritten once.  For an output of several GB (BASELINE config 5: 7.1 GB per shard) that is the
// whole prologue.  This kernel walks the partition instead (it is read for the validation anyway) and
// zeroes only
//   * the rows between the rows of consecutive groups (rows without a group: no edges),
//   * the row of a group without edges (a caller-made partition may hold one),
//   * a row that continues from the previous work item (both items add to it atomically).
// One wavefront per 64 consecutive groups; every gap is cleared by the whole wavefront (coalesced).
// The result is only meaningful for a canonical partition: the validation raises the flag as before and
// post_prologue_kernel, launched behind this kernel, then clears the whole output.
__global__ void __launch_bounds__(kBlock)
sparse_prologue_kernel(float *__restrict__ Y, int64_t N, int D, const int32_t *__restrict__ p2n,
                       const int32_t *__restrict__ pp, int64_t P, int G, int32_t *flag, int32_t seq, int validate,
                       unsigned long long *gaps, int64_t big_rows)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const bool vec_ok = (D & 3) == 0 && (reinterpret_cast<uintptr_t>(Y) & 15) == 0;
    auto clear_rows = [&](int64_t lo, int64_t cnt) {      // wave-uniform arguments
        // a long run of rows without edges (trailing isolated or padded rows of a multi-GB shard) is not one
        // wavefront's job: it goes on the call's gap list and post_prologue_kernel clears it with the whole grid
        if (gaps && cnt >= big_rows) {
            unsigned long long idx = 0;
            if (lane == 0) idx = atomicAdd(&gaps[0], 1ull);
            idx = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(idx >> 32)) << 32) |
                  (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);
            if (idx < (unsigned long long)kGapEntries) {
                if (lane == 0) { gaps[2 + 2 * idx] = (unsigned long long)lo; gaps[3 + 2 * idx] = (unsigned long long)cnt; }
                return;
            }
        }
        float *base = Y + (size_t)lo * (size_t)D;
        const size_t n = (size_t)cnt * (size_t)D;
        if (vec_ok) {
            f32x4 *b4 = reinterpret_cast<f32x4 *>(base);
            const f32x4 z = (f32x4)(0.f);
            for (size_t i = lane; i < (n >> 2); i += kWave) b4[i] = z;
        } else {
            for (size_t i = lane; i < n; i += kWave) base[i] = 0.f;
        }
    };
    if (wave == 0) {                                      // rows before the first group's row
        int64_t first = p2n[0];
        first = first < 0 ? 0 : (first > N ? N : first);
        if (first > 0) clear_rows(0, first);
    }
    bool bad = false;
    for (int64_t g0 = wave * kWave; g0 < P; g0 += nwaves * kWave) {
        const int64_t g = g0 + lane;
        const bool valid = g < P;
        int64_t lo = 0, cnt = 0;
        if (valid) {
            const int64_t r = p2n[g];
            const int64_t nxt = g + 1 < P ? (int64_t)p2n[g + 1] : N;
            const int a = pp[g], b = pp[g + 1];
            if (b < a || (g + 1 < P && nxt < r)) bad = true;
            const bool continues = g > 0 && (g % G) == 0 && p2n[g - 1] == r;
            const bool self = b <= a || continues;
            lo = self ? r : r + 1;
            int64_t hi = (self && nxt < r + 1) ? r + 1 : nxt;    // (the next group may belong to the same row)
            lo = lo < 0 ? 0 : lo;
            hi = hi > N ? N : hi;
            cnt = hi > lo ? hi - lo : 0;
        }
        unsigned long long m = __ballot(cnt > 0);
        while (m) {
            const int l = __builtin_ctzll(m);
            m &= m - 1;
            const int64_t lo_s = ((int64_t)__builtin_amdgcn_readlane((int)(lo >> 32), l) << 32) |
                                 (uint32_t)__builtin_amdgcn_readlane((int)lo, l);
            const int64_t cnt_s = ((int64_t)__builtin_amdgcn_readlane((int)(cnt >> 32), l) << 32) |
                                  (uint32_t)__builtin_amdgcn_readlane((int)cnt, l);
            clear_rows(lo_s, cnt_s);
        }
    }
    if (validate && bad) *flag = seq;
}

// behind a sparse prologue: the whole output when the validation found the partition not canonical (the streaming
// kernel then adds every row atomically), else the long gaps the prologue put on the call's list; returns at once
// when there is neither
__global__ void __launch_bounds__(kBlock)
post_prologue_kernel(float *__restrict__ Y, size_t n_floats, int D, const int32_t *flag, int32_t seq,
                     const unsigned long long *gaps)
{
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nthreads = (size_t)gridDim.x * blockDim.x;
    if (*flag == seq) {
        for (size_t i = tid; i < n_floats; i += nthreads) Y[i] = 0.f;
        return;
    }
    if (!gaps) return;
    const unsigned long long n = gaps[0] < (unsigned long long)kGapEntries ? gaps[0] : (unsigned long long)kGapEntries;
    for (unsigned long long k = 0; k < n; k++) {
        float *base = Y + (size_t)gaps[2 + 2 * k] * (size_t)D;
        const size_t cnt = (size_t)gaps[3 + 2 * k] * (size_t)D;
        for (size_t i = tid; i < cnt; i += nthreads) base[i] = 0.f;
    }
}

// ---- GCN pre-scaling: Xs[j, :] = deg[j] * X[j, :] -------------------------------------------
// Lets the degree-weighted aggregation run as an unweighted gather of Xs with one multiply by
// deg[i] at the flush (deg_i * sum_j deg_j x_j), instead of one extra 4-byte gather plus a
// broadcast and a multiply per edge.
__global__ void __launch_bounds__(kBlock)
scale_rows_kernel(const float *__restrict__ X, const float *__restrict__ deg, float *__restrict__ Xs,
                  int64_t rows, int D, int ld_out)
{
    // Xs[r, 0:D] = (deg ? deg[r] : 1) * X[r, 0:D], rows of Xs `ld_out` floats apart (pad never read)
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nthreads = (size_t)gridDim.x * blockDim.x;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    if (ld_out == D && (D & 3) == 0 &&
        ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Xs)) & 15) == 0) {
        const size_t n4 = ((size_t)rows * (size_t)D) >> 2;
        const int d4 = D >> 2;
        for (size_t i = tid; i < n4; i += nthreads) {
            const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(X) + i);
            reinterpret_cast<f32x4 *>(Xs)[i] = deg ? v * deg[i / d4] : v;
        }
    } else if ((D & 3) == 0 && (ld_out & 3) == 0 &&
               ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Xs)) & 15) == 0) {
        // padded / gapped layout of rows whose width is a whole number of 16-byte pieces: one vector per thread
        const size_t n4 = ((size_t)rows * (size_t)D) >> 2;
        const unsigned d4 = (unsigned)D >> 2, ld4 = (unsigned)ld_out >> 2;
        for (size_t i = tid; i < n4; i += nthreads) {
            const size_t r = i / d4;
            const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(X) + i);
            reinterpret_cast<f32x4 *>(Xs)[r * ld4 + (i - r * d4)] = deg ? v * deg[r] : v;
        }
    } else {
        const size_t total = (size_t)rows * (size_t)D;
        for (size_t i = tid; i < total; i += nthreads) {
            const size_t r = i / D;
            const float v = __builtin_nontemporal_load(X + i);
            Xs[r * (size_t)ld_out + (i - r * D)] = deg ? v * deg[r] : v;
        }
    }
}

// ---- main kernel --------------------------------------------------------------------------

template <int VEC, int LPR, int MODE, int U, bool WIDE, bool PHASED>
__global__ void __launch_bounds__(kBlock)
agg_kernel(const AggParams p)
{
    typedef typename VecOf<VEC>::T VT;
    typedef typename VecOf<VEC>::M MT;
    // byte offsets into X: 32-bit (SGPR base + VGPR offset addressing) unless X exceeds 4 GiB
    typedef typename std::conditional<WIDE, uint64_t, uint32_t>::type OffT;
    constexpr int RPI = kWave / LPR;  // neighbor rows per wave-wide load
    static_assert(U * RPI <= kWave, "a batch must fit one 64-edge id tile");

    const int lane = threadIdx.x & (kWave - 1);
    const int wib = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int slot = lane / LPR;
    const int c = lane % LPR;
    const int D = p.D;
    const int G = p.G;
    const bool canonical = p.trust || (*p.flag != p.seq);
    const char *xbase = reinterpret_cast<const char *>(p.X);
    const OffT row_bytes = (OffT)p.ldx * (OffT)sizeof(float);

    // item = 4 consecutive chunks handled by the 4 waves of one block.
    const int64_t item_span = p.xcd_remap ? p.items_per_xcd * kXcds : p.num_items;
    for (int64_t it = blockIdx.x; it < item_span; it += gridDim.x) {
        int64_t item = it;
        if (p.xcd_remap) {
            // blocks land on XCD (blockIdx % 8): give each XCD one contiguous range of items
            item = (it % kXcds) * p.items_per_xcd + it / kXcds;
            if (item >= p.num_items) continue;
        }
        const int64_t chunk = item * kWavesPerBlock + wib;
        if (chunk >= p.num_chunks) continue;
        const int64_t g0 = chunk * G;
        const int ng = (int)(p.P - g0 < (int64_t)G ? p.P - g0 : (int64_t)G);

        // chunk metadata: one coalesced load each (G <= 63)
        const int my_row = lane < ng ? p.p2n[g0 + lane] : -1;
        const int my_pp = lane <= ng ? p.pp[g0 + lane] : 0;
        int prev_row = -1, next_row = -1;
        if (g0 > 0) prev_row = p.p2n[g0 - 1];
        if (g0 + ng < p.P) next_row = p.p2n[g0 + ng];

        int up_row = __shfl_up(my_row, 1);
        bool is_start = lane < ng && (lane == 0 || my_row != up_row || !canonical);
        unsigned long long starts = __ballot(is_start);

        // column-phased schedule: every run keeps a cursor (next unconsumed edge) between the
        // phase launches, stored at the index of the run's first group
        int my_cur = 0, new_cur = 0;
        if constexpr (PHASED) {
            if (p.phase > 0 && lane < ng) my_cur = p.cursor[g0 + lane];
        }

        while (starts) {
            const int js = __builtin_ctzll(starts);
            starts &= starts - 1;
            const int je = starts ? __builtin_ctzll(starts) : ng;
            const int row = __builtin_amdgcn_readlane(my_row, js);
            int sb = __builtin_amdgcn_readlane(my_pp, js);
            const int se = __builtin_amdgcn_readlane(my_pp, je);
            const bool shared = (js == 0 && prev_row == row) || (je == ng && next_row == row);
            const bool use_atomic = shared || !canonical;
            if constexpr (PHASED) {
                if (p.phase > 0) sb = __builtin_amdgcn_readlane(my_cur, js);
                if (lane == js) new_cur = sb;
                if (sb >= se) continue;  // run already fully consumed by earlier phases
            }
            const bool accumulate = p.acc_in || (PHASED && p.phase > 0);

            float row_deg = 1.f;
            if constexpr (MODE == MODE_GCN) row_deg = p.deg_row[row];
            float row_scale = p.eps;
            if constexpr (MODE == MODE_GIN) {
                if (p.row_scale) row_scale *= p.row_scale[row];
            }

            int consumed_end = sb;
            for (int d0 = 0; d0 < D; d0 += VEC * LPR) {
                // lane c owns the VEC floats starting at `dcol`.  When D is not a multiple of VEC the
                // last piece is shifted back to end exactly at D: it overlaps its predecessor by
                // `shift` floats, which both lanes compute identically (so plain stores may
                // overlap) and which only the predecessor adds atomically.
                const int piece = d0 + c * VEC;
                const bool cvalid = piece < D;
                int dcol = piece, shift = 0;
                if (VEC > 1 && piece + VEC > D && cvalid) { dcol = D - VEC; shift = piece - dcol; }
                // lanes past the end of the row re-read this sweep's first piece (same cache lines, never
                // stored); min() keeps that read inside the row when the first piece is the shifted one
                const OffT col_off = (OffT)(cvalid ? dcol : (d0 + VEC <= D ? d0 : D - VEC)) * (OffT)sizeof(float);
                VT acc = vzero<VEC>();

                // gathers `nv` neighbor rows whose ids sit in lanes 0..nv-1 of `id`
                auto gather_tile = [&](const int id, const float dgn, const int nv) {
                    if (nv <= RPI) {
                        // short run (low-degree rows): one wave-wide load covers it
                        const int nid = __shfl(id, slot);
                        VT v0 = vzero<VEC>();
                        if (slot < nv) v0 = *reinterpret_cast<const MT *>(xbase + (OffT)((OffT)nid * row_bytes + col_off));
                        if constexpr (MODE == MODE_GCN) {
                            VT tmp = v0 * (row_deg * __shfl(dgn, slot));
                            acc += tmp;
                        } else {
                            acc += v0;
                        }
                        return;
                    }
#pragma unroll 1
                    for (int b = 0; b < nv; b += U * RPI) {
                        VT v[U];
                        int nid[U];
                        float cf[U];
#pragma unroll
                        for (int u = 0; u < U; u++) nid[u] = __shfl(id, b + u * RPI + slot);
                        if (b + U * RPI <= nv) {
                            // full batch: U unpredicated wave-wide row loads back to back
#pragma unroll
                            for (int u = 0; u < U; u++)
                                v[u] = *reinterpret_cast<const MT *>(xbase + (OffT)((OffT)nid[u] * row_bytes + col_off));
                        } else {
#pragma unroll
                            for (int u = 0; u < U; u++) {
                                v[u] = vzero<VEC>();
                                if (b + u * RPI + slot < nv)
                                    v[u] = *reinterpret_cast<const MT *>(xbase + (OffT)((OffT)nid[u] * row_bytes + col_off));
                            }
                        }
                        if constexpr (MODE == MODE_GCN) {
#pragma unroll
                            for (int u = 0; u < U; u++) cf[u] = row_deg * __shfl(dgn, b + u * RPI + slot);
                        }
#pragma unroll
                        for (int u = 0; u < U; u++) {
                            if constexpr (MODE == MODE_GCN) {
                                // reference rounds coef*x and the accumulate separately
                                // (__fmaf_rn(c, x, 0) then +=, .cu:405); built with -ffp-contract=off
                                VT tmp = v[u] * cf[u];
                                acc += tmp;
                            } else {
                                acc += v[u];
                            }
                        }
                    }
                };

                if constexpr (PHASED) {
                    // consume the run from its cursor while the ids stay below this phase's bound
                    // (ids of a CSR row are sorted, so that is one contiguous piece; if they are not,
                    // every edge is still consumed exactly once, only in a less local phase)
                    const bool last_phase = p.phase + 1 >= p.num_phases;
                    int t = sb;
                    while (t < se) {
                        const int nvalid = se - t < kWave ? se - t : kWave;
                        int id = 0x7fffffff;
                        if (lane < nvalid) id = __builtin_nontemporal_load(p.col + t + lane);
                        int take = nvalid;
                        if (!last_phase) {
                            const unsigned long long below = __ballot(lane < nvalid && id < p.phase_hi);
                            take = below == ~0ull ? kWave : __builtin_ctzll(~below);
                        }
                        float dgn = 0.f;
                        if constexpr (MODE == MODE_GCN) {
                            if (lane < take) dgn = p.deg_col[id];
                        }
                        gather_tile(id, dgn, take);
                        t += take;
                        if (take < nvalid) break;  // reached the phase boundary inside this tile
                    }
                    consumed_end = t;
                    if (consumed_end == sb) break;  // nothing of this run in this phase: no flush
                } else {
                    int id_next = 0;
                    if (sb + lane < se) id_next = __builtin_nontemporal_load(p.col + sb + lane);
                    for (int t = sb; t < se; t += kWave) {
                        const int nv = se - t < kWave ? se - t : kWave;
                        const int id = id_next;
                        // software prefetch of the next 64 column ids
                        if (t + kWave + lane < se) id_next = __builtin_nontemporal_load(p.col + t + kWave + lane);
                        float dgn = 0.f;
                        if constexpr (MODE == MODE_GCN) {
                            if (lane < nv) dgn = p.deg_col[id];
                        }
                        gather_tile(id, dgn, nv);
                    }
                }

                if constexpr (VEC == 4 && LPR <= 16) {
                    // Fold the RPI slots as a reduce-scatter: instead of summing all four components
                    // across the slots (8 lane swaps), the halves / rows exchange the components they
                    // do not keep.  permlane32_swap(x, z) leaves {x_lo, z_lo} | {x_hi, z_hi}: their
                    // sum holds x (lanes 0-31) and z (lanes 32-63) folded over lane, lane+32; the
                    // same for (y, w); permlane16_swap of the two results then leaves row r of the
                    // wave with component r.  3 swaps + 3 adds, and every lane ends with ONE float:
                    // component lane>>4 of piece lane%LPR (narrow rows: the slots sharing a 16-lane
                    // row are folded with DPP rotations afterwards).
                    float px, qy;
                    {
                        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[0]), __float_as_uint(acc[2]), false, false);
                        px = __uint_as_float(r[0]) + __uint_as_float(r[1]);
                        r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[1]), __float_as_uint(acc[3]), false, false);
                        qy = __uint_as_float(r[0]) + __uint_as_float(r[1]);
                    }
                    float val;
                    {
                        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(px), __float_as_uint(qy), false, false);
                        val = __uint_as_float(r[0]) + __uint_as_float(r[1]);
                    }
                    if constexpr (LPR <= 8) val += row_ror<8>(val);
                    if constexpr (LPR <= 4) val += row_ror<4>(val);
                    if constexpr (MODE == MODE_GIN) val *= row_scale;
                    const int k = lane >> 4;  // component held by this lane
                    if ((lane & 15) < LPR && cvalid) {
                        float *dst = p.Y + (size_t)row * D + dcol + k;
                        if (!use_atomic) {
                            if (accumulate) {
                                // earlier phases' partial (streamed: keep the X slice resident in L2); a shifted
                                // first piece of a later dimension sweep overlaps floats that the previous sweep
                                // has already read-modify-written: keep those as they are
                                const float prev = __builtin_nontemporal_load(dst);
                                val = (c == 0 && k < shift) ? prev : val + prev;
                            }
                            __builtin_nontemporal_store(val, dst);
                        } else if (k >= shift) {
                            unsafeAtomicAdd(dst, val);
                        }
                    }
                } else {
                    // fold the RPI slots; every slot then holds the row's partial sum
#pragma unroll
                    for (int k = 0; k < VEC; k++) {
                        float s = slot_reduce<LPR>(vget<VEC>(acc, k));
                        if constexpr (MODE == MODE_GIN) s *= row_scale;
                        vset<VEC>(acc, k, s);
                    }

                    if (slot == 0 && cvalid) {
                        float *dst = p.Y + (size_t)row * D + dcol;
                        if (!use_atomic) {
                            if (accumulate) {
                                const VT prev = __builtin_nontemporal_load(reinterpret_cast<const MT *>(dst));
#pragma unroll
                                for (int k = 0; k < VEC; k++) {
                                    const bool done = VEC > 1 && c == 0 && k < shift;
                                    vset<VEC>(acc, k, done ? vget<VEC>(prev, k) : vget<VEC>(acc, k) + vget<VEC>(prev, k));
                                }
                            }
                            __builtin_nontemporal_store(acc, reinterpret_cast<MT *>(dst));
                        } else {
#pragma unroll
                            for (int k = 0; k < VEC; k++)
                                if (k >= shift) unsafeAtomicAdd(dst + k, vget<VEC>(acc, k));
                        }
                    }
                }
            }
            if constexpr (PHASED) {
                if (lane == js) new_cur = consumed_end;
            }
        }
        if constexpr (PHASED) {
            if (is_start && p.phase + 1 < p.num_phases) p.cursor[g0 + lane] = new_cur;
        }
    }
}

// ---- host side ------------------------------------------------------------------------------

// Number of column phases.  The gather is bound by the L2-miss path once X is larger than the
// caches; restricting a launch to a slice of X makes the slice cache resident (MI355X, Reddit-like
// D=64, ids folded into one slice: 59.6 MB slice 2.81 ms, 7.45 MB 1.58 ms, 3.7 MB 1.34 ms per full
// pass).  Each extra phase costs a launch, a pass over the chunk descriptors and a
// read-modify-write of the touched output rows, so the measured optimum is about one phase per
// 14 MB of X (tools/sweep.py --phases, Reddit-like: D=16/32/64/128/256 -> 2/2/4/8/16 phases,
// +12/+17/+49/+64/+60 %), capped by the work per row and phase (products-like, average degree 50:
// 2 phases +6 %, 4 phases -12 %) and useless -- harmful -- when the ids of a row are already
// local (community-ordered variant: 1.38 ms single pass, 2.5 ms in 2 phases).  Locality and degree
// cannot be seen from here without a device round trip, so the automatic mode acts only on the
// hints the Decider supplies (gnna_tuning.avg_degree / nonlocal_ids); without hints: one pass.
thread_local int t_last_phases = 1;
thread_local int t_last_launches = 1;   // aggregation kernel launches of the calling thread's last call

// Average number of 128-byte lines one gathered row touches when rows of `row_bytes` bytes
// lie `stride_bytes` apart (row k starts at k * stride_bytes; the base is at least 128-aligned).
double avg_lines_per_row(int64_t row_bytes, int64_t stride_bytes)
{
    int64_t lines = 0;
    for (int k = 0; k < 32; k++) {
        const int64_t o = (k * stride_bytes) % 128;
        lines += (o + row_bytes + 127) / 128;
    }
    return (double)lines / 32.0;
}

// Row stride (in floats) of a staged copy of X that makes the gather cheaper, or `dim` if none
// does: strides of 4 / 16 / 32 floats are tried; a stride that is not a multiple of 4 floats also
// pays for misaligned dwordx4 accesses.  Measured on the Reddit-like graph: D = 41 -> 48:
// 2.28 -> 1.80 ms, D = 56 -> 64: 2.36 -> 1.86 ms (D = 40, 48, 64 are fine as they are).
int choose_row_stride(int dim)
{
    if (dim < 2 || dim > 256) return dim;
    auto cost = [&](int stride) {
        return avg_lines_per_row((int64_t)dim * 4, (int64_t)stride * 4) * ((stride & 3) ? 1.07 : 1.0);
    };
    const double plain = cost(dim);
    int best = dim;
    double best_cost = plain * 0.93;  // a staged copy has to gain at least 7 %
    const int cand[3] = {(dim + 3) / 4 * 4, (dim + 15) / 16 * 16, (dim + 31) / 32 * 32};
    for (int c : cand) {
        if (c == dim || c > 2 * dim) continue;
        const double cc = cost(c) * (1.0 + 0.02 * (double)(c - dim) / (double)dim);  // prefer less padding
        if (cc < best_cost) { best = c; best_cost = cc; }
    }
    return best;
}

// Row stride of the source rows the kernels gather from (== dim: X itself, no staged copy):
//  (b) a padded stride for widths whose rows straddle 128-byte lines or are not 16-byte aligned (choose_row_stride);
//  (c) a gap after every row: rows of one or two 128-byte lines that are re-read tens of times gather 2-6 % faster from a
//      copy in which every row starts at a multiple of twice its lines (measured, Reddit-like, kernel ms at equal phase
//      counts: D = 32 / 41 / 64 from strides of 64 / 128 / 128 floats: 0.805 -> 0.770, 1.480 -> 1.446, 1.508 -> 1.420; any
//      multiple of that stride measures the same, anything else -- 1.5 x, a stride off the line grid -- the same as no gap
//      or worse; rows of <= 64 bytes, D <= 16, share lines and lose by it: 0.701 -> 0.736; and so does any row once the
//      copy outgrows the Infinity Cache: D = 128 -> 256: 3.03 -> 3.30);
//  pad_rows: 0 automatic (both, for hot rows), 1 rule (b) always, 2 never, > 2 an explicit stride (experiments).
int pick_row_stride(const gnna_tuning &t, int dim, bool hot_rows, int64_t num_in_rows, bool whole_call)
{
    int ldx = dim;
    if (t.pad_rows == 1 || (t.pad_rows == 0 && hot_rows)) ldx = choose_row_stride(dim);
    if (t.pad_rows == 0 && hot_rows && dim > 16 && dim <= 64 && whole_call) {
        const int gapped = 2 * ((dim * 4 + 127) / 128) * 32;
        if ((size_t)num_in_rows * (size_t)gapped * sizeof(float) <= ((size_t)160 << 20)) ldx = gapped;
    }
    if (t.pad_rows > 2 && t.pad_rows >= dim) ldx = t.pad_rows;
    return ldx;
}

}  // namespace

// Number of phases of the sliced schedule from the slice statistics of the partition (st.cells[l] =
// non-empty (group, slice) cells when the S = 32 fine slices are merged into S >> l).
//  * not at all when the column ids of a row stay near the row (>= 60 % of the edges within a window of source
//    rows that fits an XCD's L2: a locality-ordered graph gathers from a small moving window already) --
//    measurable when rows and columns share one numbering; a caller's "scattered ids" hint settles it otherwise;
//  * slices of at most 8 MiB of source rows (Reddit-like graph, D = 16 / 32 / 64 / 128: best 4 / 4 / 8 / 16
//    phases; two slices are live while the chip moves from one to the next, and an XCD's L2 is 4 MiB);
//  * at least ~16 edges per (row, slice) piece, else a phase costs more in flushes than it saves in misses -- ~8 are
//    enough for the split that makes a slice fit the Infinity Cache (products-like, average degree 50, X = 627 MB: best
//    4 phases; amazon0505-like, degree 12: none).
// Phases for the sweep kernel when the library picks it by itself (gnna_tuning.sweep = 0), or 0: the streaming kernel
// runs.  Measured (DESIGN 3.1b, round 3 end): with plain id loads and 8 row loads in flight the sweep beats the streaming
// kernel by 3-4 % where (a) a destination row is 33-64 floats -- narrower rows flush cheaply, wider ones halve the rows
// the LDS accumulators hold --, (b) the rows are long (>= 300 edges on average) and few enough that a workgroup's share
// fits the accumulators in at most two sets (Reddit-like: 233 K rows of 492 edges; products-like needs 21 sets and loses
// 15 %, the 65-edge rows of an 8-rank shard's local part lose 40 %), (c) sources and destinations are the same node set
// and the source matrix is Infinity-Cache sized (the multi-GPU shapes -- 8 x the source rows, 32 phases -- lose 10 %),
// and (d) the schedule is sliced anyway.  It takes twice the streaming kernel's phase count (no flush per piece), at most 16
// and at most partSize / 4.
int sweep_auto_phases(const gnna_tuning &t, int mode, int dim, size_t x_bytes, int64_t num_out_rows, int64_t num_in_rows,
                      double edges, int B, int num_cus, bool deterministic, int part_size)
{
    if (t.sweep != 0 || deterministic || B < 2) return 0;
    if (dim <= 32 || dim > 64 || !sweep_supports(mode, dim, x_bytes)) return 0;
    if (num_in_rows != num_out_rows || x_bytes > ((size_t)160 << 20)) return 0;
    if (edges < 300.0 * (double)std::max<int64_t>(1, num_out_rows)) return 0;
    const double cap = (double)sweep_acc_rows(dim, 1) * 0.9 * (double)std::max(8, num_cus);
    if ((double)num_out_rows > 2.0 * cap) return 0;
    if (t.column_phases >= 2) return B;                 // a forced / measured phase count is taken as it is
    // a work item is 64 groups of one phase: keep it at >= ~256 edges (partSize 32: 8 phases 1.45 ms, 16 phases 1.64)
    return std::max(2, std::min(std::min(16, 2 * B), std::max(2, part_size / 4)));
}

int choose_slices(const SlicePlanStats &st, size_t x_bytes, int S, uint32_t slice_rows, int64_t num_out_rows,
                  bool square, bool hinted_scattered)
{
    if (st.groups <= 0 || st.edges <= 0) return 1;
    if (!hinted_scattered) {
        if (square) {
            // share of the edges whose source lies within the window around the destination that an XCD's L2 keeps
            // while it walks the rows in order (+-1.75 MiB of source rows); log-linear between the histogram's
            // half-octave thresholds.  Measured on Reddit-sized graphs at D = 64: share 0.9 (hidden locality) and
            // ~0.75 (after gnna_reorder_community_i32) run fastest single pass; a random labelling has ~0.06.
            const double row_bytes = (double)x_bytes / ((double)slice_rows * S);
            const double half_rows = 1.75 * 1048576.0 / row_bytes;
            double share = 0.0;
            if (half_rows >= 256.0) {
                const double pos = std::min(23.0, 2.0 * std::log2(half_rows / 256.0));
                const int k = (int)pos;
                const double lo = st.near[k], hi = st.near[std::min(23, k + 1)];
                share = (lo + (hi - lo) * (pos - k)) / st.edges;
            }
            if (share >= 0.6) return 1;
        }
        if (st.cells[0] <= 1.15 * st.groups) return 1;       // every group inside one slice
    }
    int b = 1;
    // slices of at most 8 MiB; 4 MiB for rows of at most 64 bytes (D <= 16: a flush is one 64-byte memory-side request,
    // a quarter of a 64-float row's, so finer slices pay: Reddit-like D = 16, 2 / 4 / 8 phases: 0.752 / 0.701 / 0.728 ms)
    const double row_bytes_all = (double)x_bytes / ((double)slice_rows * S);
    const size_t slice_target = row_bytes_all <= 64.0 ? ((size_t)4 << 20) : ((size_t)8 << 20);
    while (b < S && x_bytes / b > slice_target) b <<= 1;
    int lvl = 0;
    for (int t = S; t > b; t >>= 1) lvl++;
    const double rows = std::min((double)num_out_rows, st.groups);
    auto piece = [&](int l) {   // edges per (row, slice) piece: cells of one row that are adjacent merge
        return st.edges / std::max(rows, st.cells[l] - (st.groups - rows));
    };
    // Two regimes (measured, D = 64).  Slices that fit an XCD's L2 pay from ~16 edges per (row, slice) piece -- every
    // piece costs a flush of the row: Reddit-like shards of a 2- / 4- / 8-GPU job (246 / 369 / 430 remote edges per
    // row, X = 119 / 238 / 477 MB) run fastest with 12-16 / 16 / 16 slices (15-27 edges per piece).  Slices that
    // only fit the 256 MiB Infinity Cache still pay from ~8 edges per piece, because a miss there goes to HBM:
    // products-like (50 edges per row, X = 627 MB) 4 slices of 157 MB (12.5 per piece): 3.39 ms against 3.55 with 2
    // and 3.70 single pass.  A matrix no 16 slices can bring under the Infinity Cache (BASELINE config 5's shard:
    // 7.1 GB, 14.5 edges per row) stays single pass: 13.1 ms against 14.5 with two phases.
    int levels = 0;                               // cells[l] <-> S >> l slices, l = 0 .. levels - 1 (down to 2 slices)
    for (int t = S; t > 1; t >>= 1) levels++;
    levels = std::min(levels, kSliceLevels);
    while (b > 1 && lvl < levels && piece(lvl) < 16.0) { b >>= 1; lvl++; }
    if (lvl >= levels) b = 1;
    const size_t mall = (size_t)250000000;   // (of 256 MiB; products-like D = 100: four slices of 245 MB, 7.0 ms against 7.8 with two)
    if (x_bytes > mall) {
        int bm = 2, lm = levels - 1;              // fewest slices that fit the Infinity Cache (cells[levels - 1] <-> 2 slices)
        while (bm < S && x_bytes / bm > mall) { bm <<= 1; lm--; }
        if (x_bytes / bm <= mall && lm >= 0 && piece(lm) >= 8.0) b = std::max(b, bm);
    }
    return std::max(b, 1);
}

int choose_phases(const gnna_tuning &tune, size_t x_bytes, int64_t num_parts, int part_size)
{
    int b = 1;
    if (tune.column_phases >= 1) {
        b = std::min(tune.column_phases, 16);   // (chunk-walk kernel: one launch per phase)
    } else if (tune.nonlocal_ids == 1 && tune.avg_degree > 0) {
        // measured optimum on the Reddit-like graph, D = 16 / 32 / 40 / 48 / 64 / 96 / 128:
        // 2 / 2 / 3 / 3-4 / 4 / 5 / 8 phases, i.e. one phase per ~14 MB of X, two from 12 MB on
        const size_t per_phase = 14000000;
        b = (int)std::min<size_t>(8, (x_bytes + per_phase / 2) / per_phase);
        if (b < 2 && x_bytes >= 12000000) b = 2;
        // per-row work bounds the phase count: ~50 edges per row and phase; two phases already pay from an
        // average degree of ~40 (products-like, 50.5: 3.76 -> 3.55 ms; a 61-edge local part: 0.38 -> 0.36 ms)
        b = std::min(b, std::max(tune.avg_degree / 50, tune.avg_degree >= 40 ? 2 : 1));
        const int64_t est_edges = std::min<int64_t>(num_parts * (int64_t)part_size, num_parts * (int64_t)tune.avg_degree);
        while (b > 1 && est_edges / b < ((int64_t)4 << 20)) b--;
    }
    while (b > 1 && num_parts / b < 64) b--;
    return std::max(b, 1);
}

namespace {
typedef void (*AggKernel)(const AggParams);

template <int VEC, int LPR, int MODE, int U>
AggKernel pick_wide(bool wide, bool phased)
{
    if (phased) {
        if (wide) return agg_kernel<VEC, LPR, MODE, U, true, true>;
        return agg_kernel<VEC, LPR, MODE, U, false, true>;
    }
    if (wide) return agg_kernel<VEC, LPR, MODE, U, true, false>;
    return agg_kernel<VEC, LPR, MODE, U, false, false>;
}

template <int VEC, int LPR, int MODE>
AggKernel pick_u(int u, bool wide, bool phased)
{
    constexpr int RPI = kWave / LPR;
    constexpr int UMAX = kWave / RPI;  // == LPR
    if constexpr (VEC == 4 && UMAX >= 16) {
        if (u >= 16) return pick_wide<VEC, LPR, MODE, 16>(wide, phased);
    }
    if constexpr (UMAX >= 8) {
        if (u >= 8) return pick_wide<VEC, LPR, MODE, 8>(wide, phased);
    }
    if constexpr (VEC == 4 || UMAX < 8) {
        return pick_wide<VEC, LPR, MODE, 4>(wide, phased);
    } else {
        return pick_wide<VEC, LPR, MODE, 8>(wide, phased);
    }
}

template <int VEC, int MODE>
AggKernel pick_lpr(int lpr, int u, bool wide, bool phased)
{
    switch (lpr) {
    case 4: return pick_u<VEC, 4, MODE>(u, wide, phased);
    case 8: return pick_u<VEC, 8, MODE>(u, wide, phased);
    case 16: return pick_u<VEC, 16, MODE>(u, wide, phased);
    case 32: return pick_u<VEC, 32, MODE>(u, wide, phased);
    default: return pick_u<VEC, 64, MODE>(u, wide, phased);
    }
}

template <int MODE>
AggKernel pick_vec(int vec, int lpr, int u, bool wide, bool phased)
{
    if (vec == 4) return pick_lpr<4, MODE>(lpr, u, wide, phased);
    return pick_u<1, 4, MODE>(u, wide, phased);  // rows narrower than 4 floats: one lane per float
}

AggKernel pick_kernel(int mode, int vec, int lpr, int u, bool wide, bool phased)
{
    switch (mode) {
    case MODE_GCN: return pick_vec<MODE_GCN>(vec, lpr, u, wide, phased);
    case MODE_GIN: return pick_vec<MODE_GIN>(vec, lpr, u, wide, phased);
    default: return pick_vec<MODE_SAG>(vec, lpr, u, wide, phased);
    }
}

int launch_agg(int mode, const float *input, int64_t num_in_rows, const int32_t *column_index,
               const float *degrees, const float *degrees_in, float epsilon, const int32_t *part_pointers,
               const int32_t *part2Node, float *out, int64_t num_nodes, int dim, int64_t num_parts,
               int partSize, int dimWorker, int warpPerBlock, void *stream_v, bool accumulate_into_out = false,
               int num_windows = 1, int win_begin = 0, int win_end = 1)
{
    if (num_nodes < 0 || dim < 0 || num_parts < 0 || num_in_rows < 0)
        return fail(GNNA_ERR_INVALID_ARGUMENT, "negative size (num_nodes=%lld dim=%d num_parts=%lld)",
                          (long long)num_nodes, dim, (long long)num_parts);
    if (partSize <= 0 || dimWorker <= 0 || warpPerBlock <= 0)
        return fail(GNNA_ERR_INVALID_ARGUMENT,
                          "partSize, dimWorker and warpPerBlock must be positive (got %d, %d, %d)",
                          partSize, dimWorker, warpPerBlock);
    if (num_windows < 1 || num_windows > 16 || win_begin < 0 || win_end > num_windows || win_begin >= win_end)
        return fail(GNNA_ERR_INVALID_ARGUMENT, "bad source window range [%d, %d) of %d (at most 16 windows)",
                    win_begin, win_end, num_windows);
    // (the kernels carry a destination row together with two flag bits in one 32-bit register)
    if (num_nodes >= ((int64_t)1 << 29))
        return fail(GNNA_ERR_UNSUPPORTED, "%lld destination rows in one call (at most 536870911): shard the rows",
                    (long long)num_nodes);
    if (num_nodes == 0 || dim == 0) return GNNA_OK;
    if (!out || !input) return fail(GNNA_ERR_INVALID_ARGUMENT, "null feature pointer");
    if (num_parts > 0 && (!column_index || !part_pointers || !part2Node))
        return fail(GNNA_ERR_INVALID_ARGUMENT, "null index pointer");
    if (mode == MODE_GCN && (!degrees || !degrees_in))
        return fail(GNNA_ERR_INVALID_ARGUMENT, "null degrees pointer");
    if (out == input) return fail(GNNA_ERR_INVALID_ARGUMENT, "out must not alias input");

    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    DeviceState *ds = nullptr;
    int rc = get_device_state(&ds);
    if (rc != GNNA_OK) return rc;
    LaunchGuard in_flight;      // plans dropped by another thread stay allocated until this call has enqueued its kernels

    gnna_tuning tune;
    gnna_get_tuning(&tune);
    apply_graph_hints(column_index, dim, &tune);

    int32_t *flag = nullptr;
    const int32_t seq = next_call_seq(ds, &flag);
    const int prof_call = profile_acquire_call(num_parts > 0);
    profile_record(prof_call, 0, stream);

    // prologue: zero-fill + validation.  `sparse_G` > 0: the streaming kernel is about to run a single pass with
    // `sparse_G` groups per work item and stores every row it owns, so only the other rows are cleared.
    const size_t n_floats = (size_t)num_nodes * (size_t)dim;
    // set when the call reads packed ids: the (dense) prologue then compares a sample of column_index with the copy's
    const unsigned long long *chk_sum = nullptr;
    int64_t chk_n = 0;
    int32_t *stale_flag = flag + kFlagSlots;          // the second ring, same slot
    auto run_prologue = [&](int sparse_G) -> int {
        const int validate = (num_parts > 0 && !tune.trust_canonical) ? 1 : 0;
        const int zero_fill = (accumulate_into_out || win_begin > 0) ? 0 : 1;
        if (sparse_G > 0 && zero_fill && num_parts > 0) {
            int64_t blocks = (num_parts + kBlock - 1) / kBlock;
            blocks = std::max<int64_t>(1, std::min<int64_t>(blocks, (int64_t)ds->num_cus * 8));
            unsigned long long *gaps = ds->gap_lists + (size_t)((uint32_t)seq % kGapSlots) * kGapWords;
            (void)hipMemsetAsync(gaps, 0, sizeof(unsigned long long), stream);
            const int64_t big_rows = std::max<int64_t>(64, ((int64_t)1 << 20) / std::max(1, dim * 4));   // >= 1 MiB of zeros
            hipLaunchKernelGGL(sparse_prologue_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, stream, out, num_nodes, dim,
                               part2Node, part_pointers, num_parts, sparse_G, flag, seq, validate, gaps, big_rows);
            hipLaunchKernelGGL(post_prologue_kernel, dim3((unsigned)(ds->num_cus * 8)), dim3(kBlock), 0, stream, out,
                               n_floats, dim, flag, seq, gaps);
        } else {
            size_t work = std::max(n_floats / 4, (size_t)num_parts);
            int64_t blocks = (int64_t)((work + kBlock - 1) / kBlock);
            blocks = std::max<int64_t>(1, std::min<int64_t>(blocks, (int64_t)ds->num_cus * 8));
            hipLaunchKernelGGL(prologue_kernel, dim3((unsigned)blocks + (chk_sum ? 1u : 0u)), dim3(kBlock), 0, stream, out, n_floats,
                               part2Node, part_pointers, num_parts, flag, seq, validate, zero_fill,
                               chk_sum ? column_index : nullptr, chk_n, chk_sum, stale_flag);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(GNNA_ERR_HIP, "prologue launch: %s", hipGetErrorString(e));
        profile_record(prof_call, 1, stream);
        return GNNA_OK;
    };
    if (num_parts == 0) return run_prologue(0);

    // vector width / lane layout: one lane per 4 consecutive floats of a row (dword-aligned
    // dwordx4 accesses, ragged tail handled by the shifted last piece), LPR lanes per row
    const int vec = dim >= 4 ? 4 : 1;
    const int pieces = (dim + vec - 1) / vec;
    int lpr = 4;
    while (lpr < 64 && lpr < pieces) lpr <<= 1;

    AggParams p;
    p.X = input; p.col = column_index; p.deg_row = degrees; p.deg_col = degrees_in; p.pp = part_pointers; p.p2n = part2Node;
    p.Y = out; p.P = num_parts; p.D = dim; p.eps = epsilon;
    p.G = std::max(1, std::min(tune.groups_per_chunk, 63));
    p.num_chunks = (num_parts + p.G - 1) / p.G;
    p.num_items = (p.num_chunks + kWavesPerBlock - 1) / kWavesPerBlock;
    p.items_per_xcd = (p.num_items + kXcds - 1) / kXcds;
    p.xcd_remap = tune.xcd_remap ? 1 : 0;
    p.flag = flag; p.seq = seq; p.trust = tune.trust_canonical ? 1 : 0;

    int64_t span = p.xcd_remap ? p.items_per_xcd * kXcds : p.num_items;
    int64_t grid = span;
    if (tune.blocks_per_cu > 0) grid = std::min<int64_t>(grid, (int64_t)tune.blocks_per_cu * ds->num_cus);
    grid = std::max<int64_t>(1, std::min<int64_t>(grid, 0x7fffffff));
    if (p.xcd_remap && grid < span) grid = std::max<int64_t>(kXcds, grid / kXcds * kXcds);  // keep it % 8 stable

    const int64_t window_rows = (num_in_rows + num_windows - 1) / num_windows;

    // Staged copy of the source rows in library scratch, made when every source row is gathered many
    // times: (a) GCN pre-scaling -- Xs_j = deg_j X_j, so that the gather is unweighted and deg_i is
    // applied at the flush; (b) a padded row stride for widths whose rows straddle 128-byte lines or
    // are not 16-byte aligned (the class counts of GCN output layers: 41, 47, 22, 7 ...).
    const int64_t est_edges = num_parts * (int64_t)(tune.avg_degree > 0 ? std::min(partSize, tune.avg_degree) : partSize / 2 + 1);
    // (automatic staging only while the copy stays small next to the 288 GB of HBM: a staged copy of a multi-GB
    // feature matrix would silently double the resident set -- the per-edge form runs on the same kernel)
    const bool hot_rows = est_edges >= 32 * num_in_rows && (size_t)num_in_rows * (size_t)dim * sizeof(float) <= ((size_t)1 << 30);
    const bool prescale = mode == MODE_GCN && (tune.gcn_prescale == 1 || (tune.gcn_prescale == 0 && hot_rows));
    const int ldx = pick_row_stride(tune, dim, hot_rows, num_in_rows, win_begin == 0 && num_windows == 1);
    p.ldx = ldx;
    p.row_scale = nullptr;
    const size_t x_bytes = (size_t)num_in_rows * (size_t)ldx * sizeof(float);
    // what the gather can touch of it (whole 128-byte lines of every row; a gapped copy's gaps never enter a cache):
    // the size the slicing decisions go by
    const size_t foot_bytes = (size_t)num_in_rows * (size_t)std::min(ldx, (dim * 4 + 127) / 128 * 32) * sizeof(float);
    const bool wide = x_bytes > 0xffffffffull;
    if (prescale || ldx != dim) {
        void *xs = nullptr;
        rc = get_workspace(ds, stream, 1, x_bytes, &xs);
        if (rc != GNNA_OK) return rc;
        // (windowed calls stage the rows of their own windows only: later windows may not have arrived)
        const int64_t r0 = std::min<int64_t>((int64_t)win_begin * window_rows, num_in_rows);
        const int64_t r1 = std::min<int64_t>((int64_t)win_end * window_rows, num_in_rows);
        const size_t s_bytes = (size_t)(r1 - r0) * (size_t)dim * sizeof(float);
        int64_t sblocks = (int64_t)((s_bytes / 16 + kBlock - 1) / kBlock);
        sblocks = std::max<int64_t>(1, std::min<int64_t>(sblocks, (int64_t)ds->num_cus * 8));
        hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)sblocks), dim3(kBlock), 0, stream,
                           input + (size_t)r0 * dim, prescale ? degrees_in + r0 : nullptr,
                           static_cast<float *>(xs) + (size_t)r0 * ldx, r1 - r0, dim, ldx);
        hipError_t es = hipGetLastError();
        if (es != hipSuccess) return fail(GNNA_ERR_HIP, "staging launch: %s", hipGetErrorString(es));
        p.X = static_cast<const float *>(xs);
        if (prescale) {
            p.row_scale = degrees;
            p.eps = 1.f;
            mode = MODE_GIN;  // unweighted gather + per-row factor at the flush
        }
    }
    // Streaming kernel (gnna_stream.hip): rows of >= 4 floats that are not part of a windowed sequence.  Its sliced schedule is stateless and a single launch; the number of phases is
    // tune.column_phases when set (process-wide or measured per graph), otherwise chosen from the slice
    // statistics of the partition (first sight of a graph: one counting pass + one stream synchronisation).
    if (vec == 4 && num_windows == 1 && tune.stream_kernel != 2) {
        int B = 1;
        const uint8_t *cnt = nullptr;
        int S = kMaxSlices;
        SlicePlan plan;
        const bool can_slice = num_parts >= 1024 && num_in_rows >= 64 && foot_bytes >= ((size_t)2 << 20);
        if (tune.column_phases >= 2 && num_in_rows >= kMaxSlices) {
            B = std::min(tune.column_phases, kMaxSlices);
            rc = get_slice_plan(ds, stream, column_index, part_pointers, part2Node, num_parts, num_in_rows, false, false, &plan);
            if (rc != GNNA_OK) return rc;
        } else if (tune.column_phases == 0 && can_slice && foot_bytes >= ((size_t)6 << 20)) {
            rc = get_slice_plan(ds, stream, column_index, part_pointers, part2Node, num_parts, num_in_rows, true, false, &plan);
            if (rc != GNNA_OK) return rc;
            if (plan.cnt && plan.stats.valid)
                B = choose_slices(plan.stats, foot_bytes, plan.S, plan.slice_rows, num_nodes, num_in_rows == num_nodes,
                                  tune.nonlocal_ids == 1);
        }
        cnt = plan.cnt;
        if (cnt) S = plan.S;
        if (!cnt || B < 2) { B = 1; cnt = nullptr; }
        // Destination-blocked sweep (gnna_sweep.hip): the sliced schedule with the partial rows kept in LDS across
        // the slices -- no flush per (row, slice) piece, so it takes finer slices than the streaming kernel's rule.
        const int auto_Bs = (cnt && plan.stats.valid) ? sweep_auto_phases(tune, mode, dim, foot_bytes, num_nodes, num_in_rows, plan.stats.edges, B,
                                                                          ds->num_cus, tune.deterministic == 1, partSize) : 0;
        if (cnt && (tune.sweep == 1 || auto_Bs > 0) && tune.deterministic != 1 && sweep_supports(mode, dim, x_bytes)) {
            int Bs = B;
            if (auto_Bs > 0) {
                Bs = std::min(auto_Bs, S);
            } else if (tune.column_phases < 2) {      // sweep forced, phases not: twice the streaming kernel's
                Bs = std::max(2, std::min(std::min(std::min(16, 2 * B), std::max(2, partSize / 4)), S));
            }
            const int32_t *sw_ids = nullptr; const uint32_t *sw_off = nullptr;
            if (plan.handle && (tune.pack_ids == 1 || (tune.pack_ids == 0 && plan.pinned)) && tune.xcd_remap != 0) {
                rc = get_packed_ids(ds, stream, plan.handle, Bs, kWave, true, false, &sw_ids, &sw_off, &chk_sum, &chk_n);
                if (rc != GNNA_OK) return rc;
                if (sw_ids) count_event(CTR_PACKED_LAUNCHES);
            }
            rc = run_prologue(0);
            if (rc != GNNA_OK) return rc;
            uint32_t *sync = ds->sweep_sync + (size_t)((uint32_t)seq % kSweepSyncSlots) * (kXcds * 16);
            hipError_t em = hipMemsetAsync(sync, 0, kXcds * 16 * sizeof(uint32_t), stream);
            if (em != hipSuccess) return fail(GNNA_ERR_HIP, "sweep counters: %s", hipGetErrorString(em));
            SweepLaunch w;
            w.mode = mode; w.X = p.X; w.col = column_index; w.pp = part_pointers; w.p2n = part2Node; w.Y = out;
            w.cnt = cnt; w.row_scale = p.row_scale; w.flag = flag; w.seq = seq; w.trust = p.trust; w.sync = sync;
            w.P = num_parts; w.D = dim; w.ldx = ldx; w.S = S; w.B = Bs;
            w.U = std::max(tune.loads_in_flight, 8);      // 16 wavefronts per CU: more loads in flight per wavefront pay here
            w.rows_with_edges = plan.stats.valid ? (int64_t)std::min((double)num_nodes, plan.stats.groups) : num_nodes;
            if (tune.groups_per_chunk > 64) w.rounds = tune.groups_per_chunk / 64;   // experiments: G = 64 * (sets per workgroup)
            w.slack = tune.sweep_slack; w.wgs_per_cu = tune.blocks_per_cu;
            w.dynamic = tune.xcd_remap != 0;      // (experiments: XCD=0 selects the fixed shares per wavefront)   // (experiments: BPC = 1 / 2 workgroups per CU)
            w.plain_ok = !accumulate_into_out; w.eps = p.eps;
            w.ids_packed = sw_ids; w.item_off = sw_off; w.packed_stale = stale_flag;
            t_last_phases = Bs;
            t_last_launches = 1;
            rc = launch_sweep(ds, w, stream);
            if (rc != GNNA_OK) return rc;
            profile_record(prof_call, 2, stream);
            return GNNA_OK;
        }
        t_last_phases = B;
        StreamLaunch a;
        a.mode = mode; a.X = p.X; a.col = column_index; a.pp = part_pointers; a.p2n = part2Node; a.Y = out;
        a.cnt = cnt; a.row_scale = p.row_scale; a.deg_row = p.deg_row; a.deg_col = p.deg_col; a.flag = flag; a.seq = seq; a.trust = p.trust; a.P = num_parts;
        // a work item is (chunk, slice): keep its edge count about what `groups_per_chunk` groups are in one pass
        a.D = dim; a.ldx = ldx; a.G = std::min(64, tune.groups_per_chunk * B); a.U = tune.loads_in_flight; a.S = S; a.B = B;
        t_last_launches = 1;
        a.wide = wide; a.plain_ok = (B == 1 && !accumulate_into_out); a.xcd_remap = tune.xcd_remap != 0;
        a.eps = p.eps;
        if (tune.deterministic == 1) {
            // ordered phase launches, owned rows read-modify-written, rows shared between chunks summed in chunk order
            // from partials parked in the stream's scratch (slot 2): bit-reproducible for a canonical partition
            const int G_eff = std::max(1, std::min(a.G, kWave));
            const size_t chunks = (size_t)((num_parts + G_eff - 1) / G_eff);
            const size_t part_bytes = ((chunks * 2 * (size_t)dim * sizeof(float)) + 255) & ~(size_t)255;
            const size_t stamp_bytes = chunks * 2 * sizeof(int32_t);
            void *ws = nullptr;
            rc = get_workspace(ds, stream, 2, part_bytes + stamp_bytes, &ws);
            if (rc != GNNA_OK) return rc;
            a.det = true;
            a.det_part = static_cast<float *>(ws);
            a.det_stamp = reinterpret_cast<int32_t *>(static_cast<char *>(ws) + part_bytes);
            hipError_t em = hipMemsetAsync(a.det_stamp, 0, stamp_bytes, stream);
            if (em != hipSuccess) return fail(GNNA_ERR_HIP, "deterministic schedule scratch: %s", hipGetErrorString(em));
            t_last_launches = B;
        }
        // prepared graph: the ids in the order the sliced schedule consumes them (built by gnna_prepare_graph for the
        // widths it was given; a width or phase count it has not seen gets its copy at first use, outside captures)
        if (cnt && plan.handle && (tune.pack_ids == 1 || (tune.pack_ids == 0 && plan.pinned)) && mode != MODE_SDDMM) {
            rc = get_packed_ids(ds, stream, plan.handle, B, std::max(1, std::min(a.G, kWave)), true, false, &a.ids_packed, &a.item_off,
                                &chk_sum, &chk_n);
            a.packed_stale = stale_flag;
            if (rc != GNNA_OK) return rc;
            if (a.ids_packed) count_event(CTR_PACKED_LAUNCHES);
        }
        // single pass, nothing to add to: only the rows the kernel does not store need clearing -- worth a second
        // (empty) launch once the output is tens of MB
        const bool sparse = !a.det && a.plain_ok && tune.zero_fill != 2 &&
                            (tune.zero_fill == 1 || n_floats * sizeof(float) >= ((size_t)32 << 20));
        rc = run_prologue(sparse ? std::max(1, std::min(a.G, kWave)) : 0);
        if (rc != GNNA_OK) return rc;
        rc = launch_stream(a, stream);
        if (rc != GNNA_OK) return rc;
        profile_record(prof_call, 2, stream);
        return GNNA_OK;
    }
    rc = run_prologue(0);
    if (rc != GNNA_OK) return rc;

    // phases: `sub` launches per source window (one window == the whole source range unless the caller
    // pipelines a chunked feature exchange); this call runs the launches of windows [win_begin, win_end)
    const int total = choose_phases(tune, x_bytes, num_parts, partSize);
    const int sub = std::max(1, (total + num_windows / 2) / num_windows);
    const int phases = sub * num_windows;
    t_last_phases = phases;
    t_last_launches = sub * (win_end - win_begin);
    AggKernel k = pick_kernel(mode, vec, lpr, tune.loads_in_flight, wide, phases > 1);
    p.cursor = nullptr; p.phase = 0; p.num_phases = 1; p.phase_hi = 0x7fffffff;
    p.acc_in = accumulate_into_out ? 1 : 0;
    if (phases > 1) {
        rc = claim_cursors(ds, stream, column_index, part_pointers, num_windows, win_begin, win_end);
        if (rc != GNNA_OK) return rc;
        void *ws = nullptr;
        rc = get_workspace(ds, stream, 0, (size_t)num_parts * sizeof(int32_t), &ws);
        if (rc != GNNA_OK) return rc;
        p.cursor = static_cast<int32_t *>(ws);
        p.num_phases = phases;
    }
    const int64_t sub_rows = (window_rows + sub - 1) / sub;
    for (int w = win_begin; w < win_end; w++) {
        for (int j = 0; j < sub; j++) {
            p.phase = w * sub + j;
            const int64_t hi = std::min<int64_t>((int64_t)w * window_rows + (int64_t)(j + 1) * sub_rows,
                                                 (int64_t)(w + 1) * window_rows);
            p.phase_hi = (int32_t)std::min<int64_t>(hi, 0x7fffffff);
            hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(kBlock), 0, stream, p);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return fail(GNNA_ERR_HIP, "aggregation launch: %s", hipGetErrorString(e));
        }
    }
    profile_record(prof_call, 2, stream);
    return GNNA_OK;
}

}  // namespace
}  // namespace gnna

using namespace gnna;

extern "C" {
#pragma GCC visibility push(default)

int gnna_sag_f32(const float *input, const int32_t *row_pointers, const int32_t *column_index,
                 const float *degrees, const int32_t *part_pointers, const int32_t *part2Node,
                 float *out, int64_t num_nodes, int dim, int64_t num_parts,
                 int partSize, int dimWorker, int warpPerBlock, void *stream)
{
    (void)row_pointers; (void)degrees;  // unused by the reference kernel as well (.cu:186-259)
    return launch_agg(MODE_SAG, input, num_nodes, column_index, nullptr, nullptr, 1.f, part_pointers,
                      part2Node, out, num_nodes, dim, num_parts, partSize, dimWorker, warpPerBlock, stream);
}

int gnna_agg_gcn_f32(const float *input, const int32_t *row_pointers, const int32_t *column_index,
                     const float *degrees, const int32_t *part_pointers, const int32_t *part2Node,
                     float *out, int64_t num_nodes, int dim, int64_t num_parts,
                     int partSize, int dimWorker, int warpPerBlock, void *stream)
{
    (void)row_pointers;
    return launch_agg(MODE_GCN, input, num_nodes, column_index, degrees, degrees, 1.f, part_pointers,
                      part2Node, out, num_nodes, dim, num_parts, partSize, dimWorker, warpPerBlock, stream);
}

int gnna_agg_gin_f32(const float *input, const int32_t *row_pointers, const int32_t *column_index,
                     float epsilon, const int32_t *part_pointers, const int32_t *part2Node,
                     float *out, int64_t num_nodes, int dim, int64_t num_parts,
                     int partSize, int dimWorker, int warpPerBlock, void *stream)
{
    (void)row_pointers;
    return launch_agg(MODE_GIN, input, num_nodes, column_index, nullptr, nullptr, epsilon, part_pointers,
                      part2Node, out, num_nodes, dim, num_parts, partSize, dimWorker, warpPerBlock, stream);
}

int gnna_agg_rect_f32(int mode, const float *input, int64_t num_in_rows, const int32_t *column_index,
                      const float *degrees_out, const float *degrees_in, float epsilon,
                      const int32_t *part_pointers, const int32_t *part2Node, float *out,
                      int64_t num_out_rows, int dim, int64_t num_parts, int partSize, int accumulate,
                      void *stream)
{
    if (mode != MODE_SAG && mode != MODE_GCN && mode != MODE_GIN)
        return fail(GNNA_ERR_INVALID_ARGUMENT, "unknown mode %d", mode);
    return launch_agg(mode, input, num_in_rows, column_index, degrees_out, degrees_in, epsilon, part_pointers,
                      part2Node, out, num_out_rows, dim, num_parts, partSize, 32, 4, stream, accumulate != 0);
}

int gnna_agg_rect_windows_f32(int mode, const float *input, int64_t num_in_rows, const int32_t *column_index,
                              const float *degrees_out, const float *degrees_in, float epsilon,
                              const int32_t *part_pointers, const int32_t *part2Node, float *out,
                              int64_t num_out_rows, int dim, int64_t num_parts, int partSize, int accumulate,
                              int num_windows, int window_begin, int window_end, void *stream)
{
    if (mode != MODE_SAG && mode != MODE_GCN && mode != MODE_GIN)
        return fail(GNNA_ERR_INVALID_ARGUMENT, "unknown mode %d", mode);
    return launch_agg(mode, input, num_in_rows, column_index, degrees_out, degrees_in, epsilon, part_pointers,
                      part2Node, out, num_out_rows, dim, num_parts, partSize, 32, 4, stream, accumulate != 0,
                      num_windows, window_begin, window_end);
}

int gnna_prepare_graph(const int32_t *column_index, const int32_t *part_pointers, const int32_t *part2Node,
                       int64_t num_parts, int64_t num_in_rows, int64_t num_out_rows, int partSize,
                       const int *dims, int num_dims, int *phases_out, void *stream_v)
{
    if (num_parts < 0 || num_in_rows < 0 || num_out_rows < 0 || partSize <= 0 || num_dims < 0 || (num_dims > 0 && !dims))
        return fail(GNNA_ERR_INVALID_ARGUMENT, "gnna_prepare_graph: bad argument");
    for (int i = 0; i < num_dims; i++)
        if (phases_out) phases_out[i] = 1;
    if (num_parts == 0) return GNNA_OK;
    if (!column_index || !part_pointers || !part2Node) return fail(GNNA_ERR_INVALID_ARGUMENT, "null index pointer");
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(stream, &cap);
    if (cap != hipStreamCaptureStatusNone)
        return fail(GNNA_ERR_UNSUPPORTED, "gnna_prepare_graph synchronises and allocates: call it before the capture");
    DeviceState *ds = nullptr;
    int rc = get_device_state(&ds);
    if (rc != GNNA_OK) return rc;
    drain_dead_buffers();       // (this call may synchronise and allocate: the place to free what finalizers left behind)
    SlicePlan plan;
    rc = get_slice_plan(ds, stream, column_index, part_pointers, part2Node, num_parts, num_in_rows, true, true, &plan);
    if (rc != GNNA_OK) return rc;
    if (!plan.cnt || !plan.stats.valid) return fail(GNNA_ERR_HIP, "gnna_prepare_graph: the slice plan could not be built");
    gnna_tuning tune;
    gnna_get_tuning(&tune);
    const bool hot = plan.stats.edges >= 32.0 * (double)num_in_rows;
    size_t staged = 0, det_bytes = 0;
    for (int i = 0; i < num_dims; i++) {
        const int dim = dims[i];
        if (dim <= 0) return fail(GNNA_ERR_INVALID_ARGUMENT, "gnna_prepare_graph: dims[%d] = %d", i, dim);
        gnna_tuning t = tune;
        apply_graph_hints(column_index, dim, &t);
        const size_t raw = (size_t)num_in_rows * (size_t)dim * sizeof(float);
        const bool hot_rows = hot && raw <= ((size_t)1 << 30);
        const int ldx = pick_row_stride(t, dim, hot_rows, num_in_rows, true);
        const size_t x_bytes = (size_t)num_in_rows * (size_t)ldx * sizeof(float);
        const size_t foot_bytes = (size_t)num_in_rows * (size_t)std::min(ldx, (dim * 4 + 127) / 128 * 32) * sizeof(float);
        if (hot_rows || t.gcn_prescale == 1 || ldx != dim) staged = std::max(staged, x_bytes);   // (GCN pre-scaling stages too)
        int B = 1;
        const bool can_slice = num_parts >= 1024 && num_in_rows >= 64 && foot_bytes >= ((size_t)2 << 20);
        if (dim >= 4 && t.stream_kernel != 2) {
            if (t.column_phases >= 2 && num_in_rows >= kMaxSlices) B = std::min(t.column_phases, kMaxSlices);
            else if (t.column_phases == 0 && can_slice && foot_bytes >= ((size_t)6 << 20))
                B = choose_slices(plan.stats, foot_bytes, plan.S, plan.slice_rows, num_out_rows, num_in_rows == num_out_rows,
                                  t.nonlocal_ids == 1);
        }
        if (phases_out) phases_out[i] = std::max(1, B);
        if (B >= 2 && t.pack_ids != 2 && plan.handle) {   // (the plan is pinned here)
            const int32_t *ids = nullptr; const uint32_t *off = nullptr;
            // the kernel that will run at this width: the sweep (its own phase count, 64 groups per chunk) or the streaming kernel
            const int mode_guess = t.gcn_prescale == 2 ? MODE_GCN : MODE_SAG;
            int Bs = sweep_auto_phases(t, mode_guess, dim, foot_bytes, num_out_rows, num_in_rows, plan.stats.edges, B, ds->num_cus,
                                       t.deterministic == 1, partSize);
            if (t.sweep == 1 && t.deterministic != 1 && sweep_supports(mode_guess, dim, x_bytes))
                Bs = t.column_phases >= 2 ? B : std::max(2, std::min(std::min(16, 2 * B), std::max(2, partSize / 4)));
            Bs = std::min(Bs, plan.S);
            if (Bs >= 2 && t.xcd_remap != 0)
                rc = get_packed_ids(ds, stream, plan.handle, Bs, kWave, true, true, &ids, &off);
            else
                rc = get_packed_ids(ds, stream, plan.handle, B, std::max(1, std::min(kWave, t.groups_per_chunk * B)), true, true, &ids, &off);
            if (rc != GNNA_OK) return rc;
            if (phases_out && Bs >= 2) phases_out[i] = Bs;
        }
        if (t.deterministic == 1 && dim >= 4) {   // the deterministic schedule parks partial rows in the stream's scratch (slot 2)
            const int G_eff = std::max(1, std::min(kWave, t.groups_per_chunk * std::max(1, B)));
            const size_t chunks = (size_t)((num_parts + G_eff - 1) / G_eff);
            det_bytes = std::max(det_bytes, (((chunks * 2 * (size_t)dim * sizeof(float)) + 255) & ~(size_t)255) +
                                                chunks * 2 * sizeof(int32_t));
        }
    }
    if (det_bytes) {
        void *ws = nullptr;
        rc = get_workspace(ds, stream, 2, det_bytes, &ws);
        if (rc != GNNA_OK) return rc;
    }
    if (staged) {
        void *ws = nullptr;
        rc = get_workspace(ds, stream, 1, staged, &ws);
        if (rc != GNNA_OK) return rc;
    }
    return GNNA_OK;
}

int gnna_release_graph(const int32_t *column_index)
{
    (void)release_slice_plans(column_index, false);
    return gnna_set_graph_hints(column_index, 0, 0);
}

int gnna_forget_graph(const int32_t *column_index)
{
    if (!column_index) return fail(GNNA_ERR_INVALID_ARGUMENT, "gnna_forget_graph: null (gnna_release_graph(NULL) drops everything)");
    (void)release_slice_plans(column_index, true);
    return gnna_set_graph_hints(column_index, 0, 0);
}

int gnna_last_num_phases(void) { return t_last_phases; }
int gnna_last_num_launches(void) { return t_last_launches; }

#pragma GCC visibility pop
}  // extern "C"
