// gnna_agg.hip -- the aggregation entry points of the C ABI (include/gnna.h) and what surrounds the two
// aggregation kernels: prologue (zero-fill + partition validation), row staging / GCN pre-scaling, the choice of the
// schedule (phases of the sliced schedule, streaming or sweep kernel), the graph lifecycle.  CDNA4 / gfx950 only.
//
// Replaces the host launchers of the reference (GNNAdvisor/GNNConv/GNNAdvisor_kernel.cu: SAG_cuda :110-184, the
// aggregation halves of spmm_forward_cuda :282-322, spmm_backward_cuda :436-470, spmm_forward_cuda_gin :575-603,
// spmm_backward_cuda_gin :712-744).  The kernels themselves: stream_kernel (gnna_stream.hip) and sweep_kernel
// (gnna_sweep.hip); the chunk-walk kernel of round 1 that used to live here was retired in 0.4.0 (rows narrower than 4
// floats and the windowed entry run on stream_kernel now).
//
// Shape of the computation (all modes):
//   out[part2Node[p], :] (+)= coef * sum_{e in [partPtr[p], partPtr[p+1])} X[colidx[e], :]
//   * a prologue kernel zero-fills `out` and checks that the partition is canonical (part2Node and partPtr
//     non-decreasing); if it is not, every group is flushed with atomics, which is correct for any partition;
//   * row staging: for widths whose rows straddle 128-byte lines (41, 47, 56 ...), for hot rows of one or two lines
//     (every row on its own 256- / 512-byte boundary) and for the pre-scaled GCN form the source rows are first copied
//     into library scratch with a line-friendly row stride (and multiplied by their degree norm); the kernels take the
//     stride as `ldx`.  A caller that hands over such a layout itself (gnna_agg_ld_f32, ld_in) is gathered from directly.
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

// ---- prologue: zero-fill + partition validation ------------------------------------------

__global__ void __launch_bounds__(kBlock)
prologue_kernel(float *__restrict__ Y, size_t n_floats, int D, int ldy, const int32_t *__restrict__ p2n,
                const int32_t *__restrict__ pp, int64_t P, int32_t *flag, int32_t seq, int validate, int zero_fill,
                const int32_t *__restrict__ chk_ids, int64_t chk_n, const unsigned long long *__restrict__ chk_sum,
                int32_t *stale_flag, uint32_t *__restrict__ clear_words, int n_clear)
{
    // the step counters of the sweep kernel behind this launch (<= kBlock words): cleared here instead of by a launch of their own
    if (blockIdx.x == 0 && (int)threadIdx.x < n_clear) clear_words[threadIdx.x] = 0u;
    // packed ids of a prepared graph: does column_index still look like the array the copy was made from?  (1024 samples:
    // catches a buffer that was rewritten, which is what happens when a promise of immutability is broken by accident.)
    // The launcher adds one block for it, so that the zero-fill does not wait behind the sampled loads.
    const unsigned fill_blocks = chk_ids ? gridDim.x - 1 : gridDim.x;
    if (blockIdx.x >= fill_blocks) {
        if (threadIdx.x < kWave) {
            const unsigned long long then = chk_sum[0], distrusted = chk_sum[4];   // [4]: a full hash differed earlier (gnna_stream.hip)
            const unsigned long long now = graph_checksum(chk_ids, chk_n, pp, P, (int)threadIdx.x);
            if (threadIdx.x == 0 && (now != then || distrusted != 0ull)) *stale_flag = seq;
        }
        return;
    }
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nthreads = (size_t)fill_blocks * blockDim.x;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    if (!zero_fill) {
        // accumulate mode: Y keeps its contents
    } else if (ldy != D) {
        // rows `ldy` floats apart (the caller's leading dimension): only the D floats of every row are the library's
        if ((D & 3) == 0 && (ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(Y) & 15) == 0) {
            const size_t n4 = n_floats >> 2;
            const unsigned d4 = (unsigned)D >> 2, ld4 = (unsigned)ldy >> 2;
            const f32x4 z = (f32x4)(0.f);
            for (size_t i = tid; i < n4; i += nthreads) {
                const size_t r = i / d4;
                reinterpret_cast<f32x4 *>(Y)[r * ld4 + (i - r * d4)] = z;
            }
        } else {
            for (size_t i = tid; i < n_floats; i += nthreads) {
                const size_t r = i / (unsigned)D;
                Y[r * (size_t)ldy + (i - r * (unsigned)D)] = 0.f;
            }
        }
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
sparse_prologue_kernel(float *__restrict__ Y, int64_t N, int D, int ldy, const int32_t *__restrict__ p2n,
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
        if (ldy != D) {                                   // rows `ldy` floats apart: row by row
            for (int64_t r = lo; r < lo + cnt; r++) {
                float *row = Y + (size_t)r * (size_t)ldy;
                for (int i = lane; i < D; i += kWave) row[i] = 0.f;
            }
            return;
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
post_prologue_kernel(float *__restrict__ Y, size_t n_floats, int D, int ldy, const int32_t *flag, int32_t seq,
                     const unsigned long long *gaps)
{
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nthreads = (size_t)gridDim.x * blockDim.x;
    // element i of a [rows, D] block whose rows lie `ldy` floats apart
    auto at = [&](float *base, size_t i) -> float * {
        if (ldy == D) return base + i;
        const size_t r = i / (unsigned)D;
        return base + r * (size_t)ldy + (i - r * (unsigned)D);
    };
    if (*flag == seq) {
        for (size_t i = tid; i < n_floats; i += nthreads) *at(Y, i) = 0.f;
        return;
    }
    if (!gaps) return;
    const unsigned long long n = gaps[0] < (unsigned long long)kGapEntries ? gaps[0] : (unsigned long long)kGapEntries;
    for (unsigned long long k = 0; k < n; k++) {
        float *base = Y + (size_t)gaps[2 + 2 * k] * (size_t)ldy;
        const size_t cnt = (size_t)gaps[3 + 2 * k] * (size_t)D;
        for (size_t i = tid; i < cnt; i += nthreads) *at(base, i) = 0.f;
    }
}

// ---- GCN pre-scaling: Xs[j, :] = deg[j] * X[j, :] -------------------------------------------
// Lets the degree-weighted aggregation run as an unweighted gather of Xs with one multiply by
// deg[i] at the flush (deg_i * sum_j deg_j x_j), instead of one extra 4-byte gather plus a
// broadcast and a multiply per edge.
__global__ void __launch_bounds__(kBlock)
scale_rows_kernel(const float *__restrict__ X, int64_t ld_in, const float *__restrict__ deg, float *__restrict__ Xs,
                  int64_t rows, int D, int ld_out)
{
    // Xs[r, 0:D] = (deg ? deg[r] : 1) * X[r, 0:D]; rows of X `ld_in`, rows of Xs `ld_out` floats apart (pads never read,
    // except by the 4-float loads of rows narrower than 4 floats, whose padding lanes are never stored)
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nthreads = (size_t)gridDim.x * blockDim.x;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const bool aligned = ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Xs)) & 15) == 0;
    if (ld_out == D && ld_in == D && (D & 3) == 0 && aligned) {
        const size_t n4 = ((size_t)rows * (size_t)D) >> 2;
        const int d4 = D >> 2;
        for (size_t i = tid; i < n4; i += nthreads) {
            const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(X) + i);
            reinterpret_cast<f32x4 *>(Xs)[i] = deg ? v * deg[i / d4] : v;
        }
    } else if ((D & 3) == 0 && (ld_out & 3) == 0 && (ld_in & 3) == 0 && aligned) {
        // padded / gapped layouts of rows whose width is a whole number of 16-byte pieces: one vector per thread
        const size_t n4 = ((size_t)rows * (size_t)D) >> 2;
        const unsigned d4 = (unsigned)D >> 2;
        const size_t ldo4 = (size_t)ld_out >> 2, ldi4 = (size_t)ld_in >> 2;
        for (size_t i = tid; i < n4; i += nthreads) {
            const size_t r = i / d4, c4 = i - r * d4;
            const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(X) + r * ldi4 + c4);
            reinterpret_cast<f32x4 *>(Xs)[r * ldo4 + c4] = deg ? v * deg[r] : v;
        }
    } else {
        const size_t total = (size_t)rows * (size_t)D;
        for (size_t i = tid; i < total; i += nthreads) {
            const size_t r = i / (unsigned)D, cc = i - r * (unsigned)D;
            const float v = __builtin_nontemporal_load(X + r * (size_t)ld_in + cc);
            Xs[r * (size_t)ld_out + cc] = deg ? v * deg[r] : v;
        }
    }
}

// ---- host side ------------------------------------------------------------------------------

// What a gather can touch of a source matrix of `num_in_rows` rows stored `ldx` floats apart: whole 128-byte lines of every
// row (a gapped copy's gaps never enter a cache) -- the size every slicing decision goes by.  ONE formula for the launch and
// for gnna_prepare_graph, so that the (phases, groups per chunk) pair prepared is the pair the launch looks up.
static size_t gather_footprint_bytes(int64_t num_in_rows, int ldx, int dim)
{
    return (size_t)num_in_rows * (size_t)std::min(ldx, (std::max(dim, 4) * 4 + 127) / 128 * 32) * sizeof(float);
}

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

// The staged-copy decision of the aggregation for a caller outside this file (gnna_sddmm.hip): where should `dim`-float source
// rows stored `ld_in` floats apart be gathered from when every row is gathered about est_edges / num_in_rows times -- from
// `input` itself (*X = input, *ldx = ld_in) or from a copy in the stream's scratch (slot 1) with the row stride
// pick_row_stride() names (rows of 17..64 floats that are re-read tens of times: every row on its own 256- / 512-byte
// boundary; widths that straddle 128-byte lines: padded), made here by one pass of scale_rows_kernel.
int stage_rows_for_gather(DeviceState *ds, hipStream_t stream, const gnna_tuning &tune, const float *input, int64_t ld_in,
                          int64_t num_in_rows, int dim, int64_t est_edges, const float **X, int *ldx_out)
{
    *X = input; *ldx_out = (int)ld_in;
    const bool hot_rows = est_edges >= 32 * num_in_rows && (size_t)num_in_rows * (size_t)dim * sizeof(float) <= ((size_t)1 << 30);
    if (dim < 4 || !hot_rows) return GNNA_OK;
    const int want = pick_row_stride(tune, dim, hot_rows, num_in_rows, true);
    if (want == dim || ld_in == want) return GNNA_OK;
    const bool aligned16 = (reinterpret_cast<uintptr_t>(input) & 15) == 0 && (ld_in & 3) == 0;
    const int gapped = 2 * ((dim * 4 + 127) / 128) * 32;
    bool direct;
    if (want == gapped && dim > 16 && dim <= 64)
        direct = (ld_in % gapped) == 0 && (reinterpret_cast<uintptr_t>(input) % ((size_t)gapped * 4)) == 0;
    else
        direct = aligned16 && avg_lines_per_row((int64_t)dim * 4, ld_in * 4) <= avg_lines_per_row((int64_t)dim * 4, (int64_t)want * 4) + 1e-9;
    if (direct) return GNNA_OK;
    const size_t x_bytes = (size_t)num_in_rows * (size_t)want * sizeof(float);
    void *xs = nullptr;
    int rc = get_workspace(ds, stream, 1, x_bytes, &xs);
    if (rc != GNNA_OK) return rc;
    const size_t s_bytes = (size_t)num_in_rows * (size_t)dim * sizeof(float);
    int64_t sblocks = (int64_t)((s_bytes / 16 + kBlock - 1) / kBlock);
    sblocks = std::max<int64_t>(1, std::min<int64_t>(sblocks, (int64_t)ds->num_cus * 8));
    hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)sblocks), dim3(kBlock), 0, stream, input, ld_in, (const float *)nullptr,
                       static_cast<float *>(xs), num_in_rows, dim, want);
    hipError_t es = hipGetLastError();
    if (es != hipSuccess) return fail(GNNA_ERR_HIP, "staging launch: %s", hipGetErrorString(es));
    *X = static_cast<const float *>(xs); *ldx_out = want;
    return GNNA_OK;
}

// Share of the edges whose source lies within `half_rows` rows of the destination row (log-linear between the half-octave
// thresholds of the counting pass' histogram).
static double near_share(const SlicePlanStats &st, double half_rows)
{
    if (st.edges <= 0 || half_rows < 256.0) return 0.0;
    const double pos = std::min(23.0, 2.0 * std::log2(half_rows / 256.0));
    const int k = (int)pos;
    const double lo = st.near[k], hi = st.near[std::min(23, k + 1)];
    return (lo + (hi - lo) * (pos - k)) / st.edges;
}

// Number of phases of the sliced schedule from the slice statistics of the partition (st.cells[l] =
// non-empty (group, slice) cells when the S = 32 fine slices are merged into S >> l).
//  * not at all when the column ids of a row stay near the row (>= 60 % of the edges -- 75 % for long rows over an
//    Infinity-Cache-resident matrix, round 5 -- within a window of source rows that fits an XCD's L2: a locality-ordered
//    graph gathers from a small moving window already) --
//    measurable when rows and columns share one numbering; a caller's "scattered ids" hint settles it otherwise;
//  * slices of at most 8 MiB of source rows (Reddit-like graph, D = 16 / 32 / 64 / 128: best 4 / 4 / 8 / 16
//    phases; two slices are live while the chip moves from one to the next, and an XCD's L2 is 4 MiB);
//  * at least ~12 edges per (row, slice) piece, else a phase costs more in flushes than it saves in misses -- ~8 are
//    enough for the split that makes a slice fit the Infinity Cache (products-like, average degree 50, X = 627 MB: best
//    4 phases; amazon0505-like, degree 12: none).
// Phases for the sweep kernel when the library picks it by itself (gnna_tuning.sweep = 0), or 0: the streaming kernel
// runs.  Measured (DESIGN.md 3, round 3 end): with plain id loads and 8 row loads in flight the sweep beats the streaming
// kernel by 3-4 % where (a) a destination row is 33-64 floats -- narrower rows flush cheaply, wider ones halve the rows
// the LDS accumulators hold --, (b) the rows are long (>= 300 edges on average) and few enough that a workgroup's share
// fits the accumulators in at most two sets (Reddit-like: 233 K rows of 492 edges; products-like needs 21 sets and loses
// 15 %, the 65-edge rows of an 8-rank shard's local part lose 40 %), (c) sources and destinations are the same node set
// and the source matrix is Infinity-Cache sized (the multi-GPU shapes -- 8 x the source rows, 32 phases -- lose 10 %),
// (d) the schedule is sliced anyway, and (e) the column ids are scattered (see below).  It takes twice the streaming kernel's
// phase count (no flush per piece), at most 16 and at most partSize / 4.
int sweep_auto_phases(const gnna_tuning &t, int mode, int dim, size_t x_bytes, int64_t num_out_rows, int64_t num_in_rows,
                      const SlicePlanStats &st, int B, int num_cus, bool deterministic, int part_size)
{
    const double edges = st.edges;
    if (t.sweep != 0 || deterministic || B < 2) return 0;
    if (dim <= 32 || dim > 64 || !sweep_supports(mode, dim, x_bytes)) return 0;
    if (num_in_rows != num_out_rows || x_bytes > ((size_t)160 << 20)) return 0;
    if (edges < 300.0 * (double)std::max<int64_t>(1, num_out_rows)) return 0;
    const double cap = (double)sweep_acc_rows(dim, 1) * 0.9 * (double)std::max(8, num_cus);
    if ((double)num_out_rows > 2.0 * cap) return 0;
    // (e) the ids are SCATTERED: every destination row has about the same share of its edges in every source slice.  In a
    // locality order (a renumbered graph, a dataset that ships community-ordered) a set's edges sit in the one or two slices
    // around its own rows: the workgroups of an XCD -- whose sets are neighbours -- then all run their heavy items in the same
    // steps on 2 slices' worth of rows, and the lock-step walk that keeps ONE slice in the L2 is what hurts.  Measured, round 5
    // (profiles/r5/sweep_vs_stream_on_locality_orders_before_fix.log, Reddit-like with hidden locality, D = 64): ids scrambled
    // sweep 1.33 ms / streaming 1.42; Rabbit order 1.94 / 1.34; the library's own renumbering and the planted order 1.91-1.94
    // against 1.16-1.19 single pass.  Share of the edges within one sweep slice (1/16 of the rows) of their destination: 0.13
    // scrambled (uniform: 1/8), 0.75-0.89 for the three locality orders.
    if (near_share(st, (double)num_in_rows / 16.0) > 0.30) return 0;
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
            // Round 5 (profiles/r5/sweep_and_stream_over_locality.log): long rows over an Infinity-Cache-resident matrix
            // (Reddit-like, 492 edges per row, 60 MB) keep paying for slices up to a share of ~0.77 -- share 0.67: single pass
            // 1.60 ms, 8 slices 1.39; share 0.81: 1.23 against 1.32 -- because a long row's flush per slice is cheap and a
            // miss of the L2 still costs a trip to the Infinity Cache; short rows over an HBM-resident matrix (products-like,
            // 49 edges per row, 627 MB) at a share of 0.62 run 2.06 ms single pass against 2.40 with two slices.
            // Not for rows of exactly one 128-byte line (D = 17 .. 32): at a share of 0.69 they run 0.699 ms single pass against
            // 0.740 with four slices (one line per edge to save, a whole line to flush per piece), while D = 16 (half a line
            // to flush) gains 4 % from the slices and D = 128 (two 64-float blocks) 9 % (`…/locality_threshold_other_widths.log`).
            const bool long_rows_cached = st.edges >= 200.0 * (double)std::max<int64_t>(1, num_out_rows) && x_bytes <= (size_t)250000000;
            const bool one_line_rows = row_bytes > 64.0 && row_bytes <= 128.0;
            if (near_share(st, half_rows) >= (long_rows_cached && !one_line_rows ? 0.75 : 0.6)) return 1;
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
    // Two regimes (measured, D = 64).  Slices that fit an XCD's L2 pay from ~12 edges per (row, slice) piece (16 until the
    // regret test of round 4: a scrambled community graph of 105 edges per row runs 7 % faster at D = 64 with 8 slices of 13
    // edges per piece than with 4 -- and 10 % SLOWER at D = 32, where a row is one line: 24 there; tests/test_regret_gpu.py,
    // profiles/r4/regret_table.log) -- every
    // piece costs a flush of the row: Reddit-like shards of a 2- / 4- / 8-GPU job (246 / 369 / 430 remote edges per
    // row, X = 119 / 238 / 477 MB) run fastest with 12-16 / 16 / 16 slices (15-27 edges per piece).  Slices that
    // only fit the 256 MiB Infinity Cache still pay from ~8 edges per piece, because a miss there goes to HBM:
    // products-like (50 edges per row, X = 627 MB) 4 slices of 157 MB (12.5 per piece): 3.39 ms against 3.55 with 2
    // and 3.70 single pass.  A matrix no 16 slices can bring under the Infinity Cache (BASELINE config 5's shard:
    // 7.1 GB, 14.5 edges per row) stays single pass: 13.1 ms against 14.5 with two phases.
    int levels = 0;                               // cells[l] <-> S >> l slices, l = 0 .. levels - 1 (down to 2 slices)
    for (int t = S; t > 1; t >>= 1) levels++;
    levels = std::min(levels, kSliceLevels);
    // (a row of one 128-byte line saves half as much per avoided miss as a two-line row, at the same cost per piece)
    const double min_piece = row_bytes_all <= 128.0 ? 24.0 : 12.0;
    while (b > 1 && lvl < levels && piece(lvl) < min_piece) { b >>= 1; lvl++; }
    if (lvl >= levels) b = 1;
    const size_t mall = (size_t)250000000;   // (of 256 MiB; products-like D = 100: four slices of 245 MB, 7.0 ms against 7.8 with two)
    if (x_bytes > mall) {
        int bm = 2, lm = levels - 1;              // fewest slices that fit the Infinity Cache (cells[levels - 1] <-> 2 slices)
        while (bm < S && x_bytes / bm > mall) { bm <<= 1; lm--; }
        if (x_bytes / bm <= mall && lm >= 0 && piece(lm) >= 8.0) b = std::max(b, bm);
    }
    return std::max(b, 1);
}

namespace {

// One aggregation call.  ld_in / ld_out: row strides of `input` / `out` in floats (>= dim).  flags: GNNA_ACCUMULATE,
// GNNA_EPILOGUE_RELU.  num_windows > 1: the call is one of a windowed sequence and covers source windows [win_begin, win_end).
int launch_agg(int mode, const float *input, int64_t ld_in, int64_t num_in_rows, const int32_t *column_index,
               const float *degrees, const float *degrees_in, float epsilon, const int32_t *part_pointers,
               const int32_t *part2Node, float *out, int64_t ld_out, int64_t num_nodes, int dim, int64_t num_parts,
               int partSize, int dimWorker, int warpPerBlock, void *stream_v, unsigned flags,
               int num_windows = 1, int win_begin = 0, int win_end = 1, bool profiled = true, int hint_dim = 0);

// Wide rows in column blocks.  A row of 256 floats is eight 128-byte lines: the lines a source slice touches stop
// fitting an XCD's L2 long before the slicing rule runs out of phases, and rows wider than 128 floats leave the sweep
// kernel's accumulators.  A 64-float column block of the same rows is the well-behaved two-line case again, so when every
// source row is gathered many times and the matrix is Infinity-Cache sized the call is split into ceil(dim / 64) calls over
// column blocks of `input` and `out` (leading dimensions: the blocks are aggregated in place; the ids are re-read per block,
// which is cheap).  Measured on the Reddit-like graph (prepared, partSize 128; tools/probe_blocks.py, profiles/r4/): D = 100 /
// 128 / 160 / 192 / 256: 2.98 / 3.16 / 4.58 / 5.34 / 7.25 -> 2.76 / 2.88 / 4.12 / 4.13 / 5.45 ms; D = 96 loses (2.39 -> 2.72).
// From 100 floats on when the rows are long enough that the blocks run on the sweep kernel (>= ~200 edges per row by the
// launcher's estimate), from 192 floats on otherwise (a graph of 198-edge rows ties at D = 128).
int column_blocks(const gnna_tuning &tune, int dim, int64_t num_in_rows, int64_t est_edges)
{
    if (tune.wide_blocks == 2 || dim < 72) return 1;
    if (tune.wide_blocks == 1) return (dim + 63) / 64;
    const bool hot = est_edges >= 32 * num_in_rows;
    if (!hot || (size_t)num_in_rows * (size_t)dim * sizeof(float) > ((size_t)250000000)) return 1;
    const bool long_rows = est_edges >= 200 * num_in_rows;
    if (dim < (long_rows ? 100 : 192)) return 1;
    return (dim + 63) / 64;
}

int launch_agg_blocked(int nb, int mode, const float *input, int64_t ld_in, int64_t num_in_rows, const int32_t *column_index,
                       const float *degrees, const float *degrees_in, float epsilon, const int32_t *part_pointers,
                       const int32_t *part2Node, float *out, int64_t ld_out, int64_t num_nodes, int dim, int64_t num_parts,
                       int partSize, int dimWorker, int warpPerBlock, void *stream_v, unsigned flags)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    const int prof_call = profile_acquire_call(num_parts > 0);
    profile_record(prof_call, 0, stream);
    profile_record(prof_call, 1, stream);
    const int w = ((dim + nb - 1) / nb + 3) / 4 * 4;        // block width: a multiple of 4 floats
    int launches = 0, phases = 1;
    for (int c0 = 0; c0 < dim; c0 += w) {
        const int wb = std::min(w, dim - c0);
        const int rc = launch_agg(mode, input + c0, ld_in, num_in_rows, column_index, degrees, degrees_in, epsilon, part_pointers,
                                  part2Node, out + c0, ld_out, num_nodes, wb, num_parts, partSize, dimWorker, warpPerBlock, stream_v,
                                  flags, 1, 0, 1, /*profiled=*/false, /*hint_dim=*/dim);
        if (rc != GNNA_OK) return rc;
        launches += t_last_launches;
        phases = std::max(phases, t_last_phases);
    }
    t_last_launches = launches;
    t_last_phases = phases;
    profile_record(prof_call, 2, stream);
    return GNNA_OK;
}

int launch_agg(int mode, const float *input, int64_t ld_in, int64_t num_in_rows, const int32_t *column_index,
               const float *degrees, const float *degrees_in, float epsilon, const int32_t *part_pointers,
               const int32_t *part2Node, float *out, int64_t ld_out, int64_t num_nodes, int dim, int64_t num_parts,
               int partSize, int dimWorker, int warpPerBlock, void *stream_v, unsigned flags,
               int num_windows, int win_begin, int win_end, bool profiled, int hint_dim)
{
    const bool accumulate_into_out = (flags & GNNA_ACCUMULATE) != 0;
    const bool relu = (flags & GNNA_EPILOGUE_RELU) != 0;
    if (flags & ~(unsigned)(GNNA_ACCUMULATE | GNNA_EPILOGUE_RELU))
        return fail(GNNA_ERR_INVALID_ARGUMENT, "unknown flag bits 0x%x", flags);
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
    if (ld_in < dim || ld_out < dim || ld_in >= ((int64_t)1 << 29) || ld_out >= ((int64_t)1 << 29))
        return fail(GNNA_ERR_INVALID_ARGUMENT, "row strides must be >= dim and < 2^29 floats (ld_in=%lld ld_out=%lld dim=%d)",
                    (long long)ld_in, (long long)ld_out, dim);
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
    // (a column block of a wide row: a schedule measured for the whole width -- gnna_set_graph_phases -- is the blocks')
    apply_graph_hints(column_index, hint_dim > 0 ? hint_dim : dim, &tune);

    const bool windowed = num_windows > 1;
    if (profiled && !windowed && num_parts > 0) {
        const int64_t est = num_parts * (int64_t)(tune.avg_degree > 0 ? std::min(partSize, tune.avg_degree) : partSize / 2 + 1);
        const int nb = column_blocks(tune, dim, num_in_rows, est);
        if (nb > 1)
            return launch_agg_blocked(nb, mode, input, ld_in, num_in_rows, column_index, degrees, degrees_in, epsilon, part_pointers,
                                      part2Node, out, ld_out, num_nodes, dim, num_parts, partSize, dimWorker, warpPerBlock, stream_v, flags);
    }
    int32_t *flag = nullptr;
    const int32_t seq = next_call_seq(ds, stream, &flag);
    const int prof_call = profiled ? profile_acquire_call(num_parts > 0) : -1;
    profile_record(prof_call, 0, stream);
    const int ldy = (int)ld_out;

    // prologue: zero-fill + validation.  `sparse_G` > 0: the streaming kernel is about to run a single pass with
    // `sparse_G` groups per work item and stores every row it owns, so only the other rows are cleared.
    const size_t n_floats = (size_t)num_nodes * (size_t)dim;
    // set when the call reads packed ids: the (dense) prologue then compares a sample of the graph with the copy's
    const unsigned long long *chk_sum = nullptr;
    int64_t chk_n = 0;
    int32_t *stale_flag = flag + kFlagSlots;          // the second ring, same slot
    auto run_prologue = [&](int sparse_G, uint32_t *clear_words = nullptr, int n_clear = 0) -> int {
        const int validate = (num_parts > 0 && !tune.trust_canonical) ? 1 : 0;
        const int zero_fill = (accumulate_into_out || win_begin > 0) ? 0 : 1;
        if (sparse_G > 0 && zero_fill && num_parts > 0) {
            int64_t blocks = (num_parts + kBlock - 1) / kBlock;
            blocks = std::max<int64_t>(1, std::min<int64_t>(blocks, (int64_t)ds->num_cus * 8));
            unsigned long long *gaps = ds->gap_lists + (size_t)call_block_of(ds, stream, seq) * kGapWords;
            (void)hipMemsetAsync(gaps, 0, sizeof(unsigned long long), stream);
            const int64_t big_rows = std::max<int64_t>(64, ((int64_t)1 << 20) / std::max(1, dim * 4));   // >= 1 MiB of zeros
            hipLaunchKernelGGL(sparse_prologue_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, stream, out, num_nodes, dim, ldy,
                               part2Node, part_pointers, num_parts, sparse_G, flag, seq, validate, gaps, big_rows);
            hipLaunchKernelGGL(post_prologue_kernel, dim3((unsigned)(ds->num_cus * 8)), dim3(kBlock), 0, stream, out,
                               n_floats, dim, ldy, flag, seq, gaps);
        } else {
            size_t work = std::max(n_floats / 4, (size_t)num_parts);
            int64_t blocks = (int64_t)((work + kBlock - 1) / kBlock);
            blocks = std::max<int64_t>(1, std::min<int64_t>(blocks, (int64_t)ds->num_cus * 8));
            hipLaunchKernelGGL(prologue_kernel, dim3((unsigned)blocks + (chk_sum ? 1u : 0u)), dim3(kBlock), 0, stream, out, n_floats,
                               dim, ldy, part2Node, part_pointers, num_parts, flag, seq, validate, zero_fill,
                               chk_sum ? column_index : nullptr, chk_n, chk_sum, stale_flag, clear_words, n_clear);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(GNNA_ERR_HIP, "prologue launch: %s", hipGetErrorString(e));
        profile_record(prof_call, 1, stream);
        return GNNA_OK;
    };
    if (num_parts == 0) {
        rc = run_prologue(0);
        if (rc != GNNA_OK || !relu || !accumulate_into_out) return rc;
        // (nothing to add, but the epilogue still applies to what `out` holds)
        StreamLaunch e0;
        e0.mode = mode; e0.Y = out; e0.p2n = part2Node; e0.P = 0; e0.D = dim; e0.ldy = ldy; e0.relu = true; e0.num_out_rows = num_nodes;
        e0.flag = flag; e0.seq = seq; e0.cnt = nullptr; e0.B = 1; e0.plain_ok = false; e0.G = 1;
        return launch_stream_epilogue_only(e0, stream);
    }

    const int64_t window_rows = (num_in_rows + num_windows - 1) / num_windows;

    // ---- where the rows are gathered from -------------------------------------------------------------------------
    // Either `input` itself (row stride ld_in) or a staged copy in library scratch, made when every source row is gathered
    // many times: (a) GCN pre-scaling -- Xs_j = deg_j X_j, so that the gather is unweighted and deg_i is applied at the
    // flush; (b) a padded row stride for widths whose rows straddle 128-byte lines or are not 16-byte aligned (the
    // class counts of GCN output layers: 41, 47, 22, 7 ...); (c) every row on its own 256- / 512-byte boundary; (d) rows
    // narrower than 4 floats, which the kernels gather as 4-float rows.  A caller whose own layout (ld_in) is already
    // what (b) / (c) would produce is gathered from directly: no copy per call.
    const int64_t est_edges = num_parts * (int64_t)(tune.avg_degree > 0 ? std::min(partSize, tune.avg_degree) : partSize / 2 + 1);
    // (automatic staging only while the copy stays small next to the 288 GB of HBM: a staged copy of a multi-GB
    // feature matrix would silently double the resident set -- the per-edge form runs on the same kernel)
    const bool hot_rows = est_edges >= 32 * num_in_rows && (size_t)num_in_rows * (size_t)dim * sizeof(float) <= ((size_t)1 << 30);
    const bool prescale = mode == MODE_GCN && (tune.gcn_prescale == 1 || (tune.gcn_prescale == 0 && hot_rows));
    const int want = dim < 4 ? 4 : pick_row_stride(tune, dim, hot_rows, num_in_rows, !windowed);
    bool direct = !prescale && dim >= 4;
    if (direct && want != dim && ld_in != want) {
        // is the caller's layout as good as the staged one would be?
        const bool aligned16 = (reinterpret_cast<uintptr_t>(input) & 15) == 0 && (ld_in & 3) == 0;
        const int gapped = 2 * ((dim * 4 + 127) / 128) * 32;
        if (want == gapped && dim > 16 && dim <= 64)
            direct = (ld_in % gapped) == 0 && (reinterpret_cast<uintptr_t>(input) % ((size_t)gapped * 4)) == 0;
        else
            direct = aligned16 && avg_lines_per_row((int64_t)dim * 4, ld_in * 4) <= avg_lines_per_row((int64_t)dim * 4, (int64_t)want * 4) + 1e-9;
    }
    const int ldx = direct ? (int)ld_in : want;
    const float *X = input;
    const float *row_scale = nullptr;
    float eps = epsilon;
    const size_t x_bytes = (size_t)num_in_rows * (size_t)ldx * sizeof(float);
    // what the gather can touch of it (whole 128-byte lines of every row; a gapped copy's gaps never enter a cache):
    // the size the slicing decisions go by
    const size_t foot_bytes = gather_footprint_bytes(num_in_rows, ldx, dim);
    const bool wide = x_bytes > 0xffffffffull;
    if (!direct) {
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
                           input + (size_t)r0 * (size_t)ld_in, ld_in, prescale ? degrees_in + r0 : nullptr,
                           static_cast<float *>(xs) + (size_t)r0 * ldx, r1 - r0, dim, ldx);
        hipError_t es = hipGetLastError();
        if (es != hipSuccess) return fail(GNNA_ERR_HIP, "staging launch: %s", hipGetErrorString(es));
        X = static_cast<const float *>(xs);
        if (prescale) {
            row_scale = degrees;
            eps = 1.f;
            mode = MODE_GIN;  // unweighted gather + per-row factor at the flush
        }
    }

    // ---- the schedule: phases of the sliced schedule ------------------------------------------------------------
    // tune.column_phases when set (process-wide or measured per graph), otherwise chosen from the slice statistics of
    // the partition (first sight of a graph: one counting pass + one stream synchronisation).
    int B = 1;
    const uint8_t *cnt = nullptr;
    int S = kMaxSlices;
    int win_lo = 0, win_hi = 0;
    SlicePlan plan;
    const bool can_slice = num_parts >= 1024 && num_in_rows >= 64 && foot_bytes >= ((size_t)2 << 20);
    if (windowed) {
        // the fine slices of this plan ARE the caller's source windows: the call takes, of every group, the ids of its
        // windows -- positions [cum[win_begin], cum[win_end]) -- which are exactly those ids when the group's ids are
        // sorted (checked by the counting pass: a window call must not read rows that have not arrived)
        rc = get_slice_plan(ds, stream, column_index, part_pointers, part2Node, num_parts, num_in_rows, true, false, &plan,
                            (uint32_t)std::max<int64_t>(1, window_rows));
        if (rc != GNNA_OK) return rc;
        if (!plan.cnt || !plan.stats.valid)
            return fail(GNNA_ERR_UNSUPPORTED, "the windowed aggregation needs its window counts, which cannot be built inside a "
                                              "stream capture: run the sequence once before capturing it");
        if (plan.stats.unsorted > 0)
            return fail(GNNA_ERR_UNSUPPORTED, "windowed aggregation: the column ids of %.0f neighbor-groups are not in increasing "
                                              "order (sort the ids of every row, as the loader's CSR has them)", plan.stats.unsorted);
        cnt = plan.cnt; S = plan.S; B = 1;
        win_lo = win_begin; win_hi = win_end == num_windows ? S : win_end;
    } else {
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
    }

    // ---- destination-blocked sweep (gnna_sweep.hip): the sliced schedule with the partial rows kept in LDS across
    // the slices -- no flush per (row, slice) piece, so it takes finer slices than the streaming kernel's rule.
    const int auto_Bs = (cnt && !windowed && plan.stats.valid)
                            ? sweep_auto_phases(tune, mode, dim, foot_bytes, num_nodes, num_in_rows, plan.stats, B, ds->num_cus,
                                                tune.deterministic == 1, partSize) : 0;
    if (cnt && !windowed && dim >= 4 && (tune.sweep == 1 || auto_Bs > 0) && tune.deterministic != 1 && sweep_supports(mode, dim, x_bytes)) {
        int Bs = B;
        if (auto_Bs > 0) {
            Bs = std::min(auto_Bs, S);
        } else if (tune.column_phases < 2) {      // sweep forced, phases not: twice the streaming kernel's
            Bs = std::max(2, std::min(std::min(std::min(16, 2 * B), std::max(2, partSize / 4)), S));
        }
        const int32_t *sw_ids = nullptr; const uint32_t *sw_off = nullptr;
        if (plan.handle && (tune.pack_ids == 1 || (tune.pack_ids == 0 && plan.pinned)) && tune.xcd_remap != 0) {
            rc = get_packed_ids(ds, stream, plan.handle, Bs, kWave, true, false, &sw_ids, &sw_off, &chk_sum, &chk_n, stale_flag, seq,
                                tune.ids_check_every);
            if (rc != GNNA_OK) return rc;
            if (sw_ids) count_event(CTR_PACKED_LAUNCHES);
        }
        uint32_t *sync = ds->sweep_sync + (size_t)call_block_of(ds, stream, seq) * kSweepSlotWords;
        static_assert(kXcds * 16 + 2 <= kBlock, "the prologue clears the sweep counters with one block");
        rc = run_prologue(0, sync, kXcds * 16 + 2);       // (dense prologue: it also zeroes the call's step counters and list header)
        if (rc != GNNA_OK) return rc;
        SweepLaunch w;
        w.mode = mode; w.X = X; w.col = column_index; w.pp = part_pointers; w.p2n = part2Node; w.Y = out;
        w.cnt = cnt; w.row_scale = row_scale; w.flag = flag; w.seq = seq; w.trust = tune.trust_canonical ? 1 : 0; w.sync = sync;
        w.P = num_parts; w.D = dim; w.ldx = ldx; w.ldy = ldy; w.S = S; w.B = Bs;
        w.relu = relu; w.num_out_rows = num_nodes;
        w.U = std::max(tune.loads_in_flight, 8);      // 16 wavefronts per CU: more loads in flight per wavefront pay here
        w.rows_with_edges = plan.stats.valid ? (int64_t)std::min((double)num_nodes, plan.stats.groups) : num_nodes;
        if (tune.groups_per_chunk > 64) w.rounds = tune.groups_per_chunk / 64;   // experiments: G = 64 * (sets per workgroup)
        w.slack = tune.sweep_slack; w.wgs_per_cu = tune.blocks_per_cu;          // (experiments: BPC = 1 / 2 workgroups per CU)
        w.dynamic = tune.xcd_remap != 0;                                          // (experiments: XCD=0 selects the fixed shares per wavefront)
        w.plain_ok = !accumulate_into_out; w.eps = eps;
        w.ids_packed = sw_ids; w.item_off = sw_off; w.packed_stale = stale_flag;
        t_last_phases = Bs;
        t_last_launches = 1;
        rc = launch_sweep(ds, w, stream);
        if (rc != GNNA_OK) return rc;
        profile_record(prof_call, 2, stream);
        return GNNA_OK;
    }

    // ---- streaming kernel (gnna_stream.hip) -----------------------------------------------------------------------
    t_last_phases = B;
    StreamLaunch a;
    a.mode = mode; a.X = X; a.col = column_index; a.pp = part_pointers; a.p2n = part2Node; a.Y = out;
    a.cnt = cnt; a.row_scale = row_scale; a.deg_row = degrees; a.deg_col = degrees_in; a.flag = flag; a.seq = seq;
    a.trust = tune.trust_canonical ? 1 : 0; a.P = num_parts;
    // a work item is (chunk, slice): keep its edge count about what `groups_per_chunk` groups are in one pass
    a.D = dim; a.ldx = ldx; a.ldy = ldy; a.G = std::min(64, tune.groups_per_chunk * B); a.U = tune.loads_in_flight; a.S = S; a.B = B;
    a.win_lo = win_lo; a.win_hi = win_hi; a.relu = relu; a.num_out_rows = num_nodes;
    t_last_launches = 1;
    a.wide = wide; a.plain_ok = (B == 1 && !accumulate_into_out && !windowed); a.xcd_remap = tune.xcd_remap != 0;
    a.eps = eps;
    if (tune.deterministic == 1 && !windowed) {
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
    if (cnt && !windowed && plan.handle && (tune.pack_ids == 1 || (tune.pack_ids == 0 && plan.pinned))) {
        rc = get_packed_ids(ds, stream, plan.handle, B, std::max(1, std::min(a.G, kWave)), true, false, &a.ids_packed, &a.item_off,
                            &chk_sum, &chk_n, stale_flag, seq, tune.ids_check_every);
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
    return launch_agg(MODE_SAG, input, dim, num_nodes, column_index, nullptr, nullptr, 1.f, part_pointers,
                      part2Node, out, dim, num_nodes, dim, num_parts, partSize, dimWorker, warpPerBlock, stream, 0u);
}

int gnna_agg_gcn_f32(const float *input, const int32_t *row_pointers, const int32_t *column_index,
                     const float *degrees, const int32_t *part_pointers, const int32_t *part2Node,
                     float *out, int64_t num_nodes, int dim, int64_t num_parts,
                     int partSize, int dimWorker, int warpPerBlock, void *stream)
{
    (void)row_pointers;
    return launch_agg(MODE_GCN, input, dim, num_nodes, column_index, degrees, degrees, 1.f, part_pointers,
                      part2Node, out, dim, num_nodes, dim, num_parts, partSize, dimWorker, warpPerBlock, stream, 0u);
}

int gnna_agg_gin_f32(const float *input, const int32_t *row_pointers, const int32_t *column_index,
                     float epsilon, const int32_t *part_pointers, const int32_t *part2Node,
                     float *out, int64_t num_nodes, int dim, int64_t num_parts,
                     int partSize, int dimWorker, int warpPerBlock, void *stream)
{
    (void)row_pointers;
    return launch_agg(MODE_GIN, input, dim, num_nodes, column_index, nullptr, nullptr, epsilon, part_pointers,
                      part2Node, out, dim, num_nodes, dim, num_parts, partSize, dimWorker, warpPerBlock, stream, 0u);
}

int gnna_agg_rect_f32(int mode, const float *input, int64_t num_in_rows, const int32_t *column_index,
                      const float *degrees_out, const float *degrees_in, float epsilon,
                      const int32_t *part_pointers, const int32_t *part2Node, float *out,
                      int64_t num_out_rows, int dim, int64_t num_parts, int partSize, int accumulate,
                      void *stream)
{
    if (mode != MODE_SAG && mode != MODE_GCN && mode != MODE_GIN)
        return fail(GNNA_ERR_INVALID_ARGUMENT, "unknown mode %d", mode);
    return launch_agg(mode, input, dim, num_in_rows, column_index, degrees_out, degrees_in, epsilon, part_pointers,
                      part2Node, out, dim, num_out_rows, dim, num_parts, partSize, 32, 4, stream,
                      accumulate != 0 ? GNNA_ACCUMULATE : 0u);
}

int gnna_agg_ld_f32(int mode, const float *input, int64_t ld_in, int64_t num_in_rows, const int32_t *column_index,
                    const float *degrees_out, const float *degrees_in, float epsilon,
                    const int32_t *part_pointers, const int32_t *part2Node, float *out, int64_t ld_out,
                    int64_t num_out_rows, int dim, int64_t num_parts, int partSize, unsigned flags, void *stream)
{
    if (mode != MODE_SAG && mode != MODE_GCN && mode != MODE_GIN)
        return fail(GNNA_ERR_INVALID_ARGUMENT, "unknown mode %d", mode);
    return launch_agg(mode, input, ld_in, num_in_rows, column_index, degrees_out, degrees_in, epsilon, part_pointers,
                      part2Node, out, ld_out, num_out_rows, dim, num_parts, partSize, 32, 4, stream, flags);
}

int gnna_agg_rect_windows_f32(int mode, const float *input, int64_t num_in_rows, const int32_t *column_index,
                              const float *degrees_out, const float *degrees_in, float epsilon,
                              const int32_t *part_pointers, const int32_t *part2Node, float *out,
                              int64_t num_out_rows, int dim, int64_t num_parts, int partSize, int accumulate,
                              int num_windows, int window_begin, int window_end, void *stream)
{
    if (mode != MODE_SAG && mode != MODE_GCN && mode != MODE_GIN)
        return fail(GNNA_ERR_INVALID_ARGUMENT, "unknown mode %d", mode);
    return launch_agg(mode, input, dim, num_in_rows, column_index, degrees_out, degrees_in, epsilon, part_pointers,
                      part2Node, out, dim, num_out_rows, dim, num_parts, partSize, 32, 4, stream,
                      accumulate != 0 ? GNNA_ACCUMULATE : 0u, num_windows, window_begin, window_end);
}

int64_t gnna_preferred_ld(int dim, int64_t num_in_rows, int64_t num_edges)
{
    if (dim <= 0 || num_in_rows <= 0) return dim > 0 ? dim : 0;
    gnna_tuning tune;
    gnna_get_tuning(&tune);
    const bool hot = num_edges >= 32 * num_in_rows && (size_t)num_in_rows * (size_t)dim * sizeof(float) <= ((size_t)1 << 30);
    return dim < 4 ? dim : pick_row_stride(tune, dim, hot, num_in_rows, true);
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
    // counted like a launch from here on: get_slice_plan's allocation path frees deferred buffers when "only the caller"
    // is in flight (drain_dead_locked(1)) -- without the guard another thread between its plan lookup and its kernel
    // launch would have been that one
    LaunchGuard in_flight;
    SlicePlan plan;
    rc = get_slice_plan(ds, stream, column_index, part_pointers, part2Node, num_parts, num_in_rows, true, true, &plan);
    if (rc != GNNA_OK) return rc;
    if (!plan.cnt || !plan.stats.valid) return fail(GNNA_ERR_HIP, "gnna_prepare_graph: the slice plan could not be built");
    gnna_tuning tune;
    gnna_get_tuning(&tune);
    const bool hot = plan.stats.edges >= 32.0 * (double)num_in_rows;
    size_t staged = 0, det_bytes = 0;
    // a width that will run in column blocks (column_blocks) is prepared as the blocks' widths: (full width, block width) pairs
    std::vector<std::pair<int, int>> widths;     // (index into dims / phases_out, width the kernels will see)
    for (int i = 0; i < num_dims; i++) {
        if (dims[i] <= 0) return fail(GNNA_ERR_INVALID_ARGUMENT, "gnna_prepare_graph: dims[%d] = %d", i, dims[i]);
        const int nb = column_blocks(tune, dims[i], num_in_rows, (int64_t)plan.stats.edges);
        if (nb <= 1) { widths.emplace_back(i, dims[i]); continue; }
        const int w = ((dims[i] + nb - 1) / nb + 3) / 4 * 4;
        widths.emplace_back(i, w);
        if (dims[i] % w) widths.emplace_back(i, dims[i] % w);
    }
    for (const auto &wd : widths) {
        const int i = wd.first;
        const int dim = wd.second;
        gnna_tuning t = tune;
        apply_graph_hints(column_index, dims[i], &t);
        const size_t raw = (size_t)num_in_rows * (size_t)dim * sizeof(float);
        const bool hot_rows = hot && raw <= ((size_t)1 << 30);
        const int ldx = pick_row_stride(t, dim, hot_rows, num_in_rows, true);
        const size_t x_bytes = (size_t)num_in_rows * (size_t)ldx * sizeof(float);
        const size_t foot_bytes = gather_footprint_bytes(num_in_rows, ldx, dim);     // (the launch's own formula: same (B, G) pair looked up)
        if (hot_rows || t.gcn_prescale == 1 || ldx != dim) staged = std::max(staged, x_bytes);   // (GCN pre-scaling stages too)
        int B = 1;
        const bool can_slice = num_parts >= 1024 && num_in_rows >= 64 && foot_bytes >= ((size_t)2 << 20);
        {
            if (t.column_phases >= 2 && num_in_rows >= kMaxSlices) B = std::min(t.column_phases, kMaxSlices);
            else if (t.column_phases == 0 && can_slice && foot_bytes >= ((size_t)6 << 20))
                B = choose_slices(plan.stats, foot_bytes, plan.S, plan.slice_rows, num_out_rows, num_in_rows == num_out_rows,
                                  t.nonlocal_ids == 1);
        }
        // (a width that runs in column blocks reports the largest phase count among its blocks, not the last block's)
        const bool first_of_width = &wd == &widths.front() || (&wd - 1)->first != i;
        if (phases_out) phases_out[i] = first_of_width ? std::max(1, B) : std::max(phases_out[i], std::max(1, B));
        if (B >= 2 && t.pack_ids != 2 && plan.handle) {   // (the plan is pinned here)
            const int32_t *ids = nullptr; const uint32_t *off = nullptr;
            // the kernel that will run at this width: the sweep (its own phase count, 64 groups per chunk) for the unweighted
            // and pre-scaled calls where the library picks it, the streaming kernel for everything else -- and for the
            // per-edge GCN form (gcn_prescale = 2) at any width, which the sweep does not run
            int Bs = sweep_auto_phases(t, MODE_SAG, dim, foot_bytes, num_out_rows, num_in_rows, plan.stats, B, ds->num_cus,
                                       t.deterministic == 1, partSize);
            if (t.sweep == 1 && t.deterministic != 1 && sweep_supports(MODE_SAG, dim, x_bytes))
                Bs = t.column_phases >= 2 ? B : std::max(2, std::min(std::min(16, 2 * B), std::max(2, partSize / 4)));
            Bs = std::min(Bs, plan.S);
            const bool swept = Bs >= 2 && t.xcd_remap != 0;
            if (swept) rc = get_packed_ids(ds, stream, plan.handle, Bs, kWave, true, true, &ids, &off);
            if (rc == GNNA_OK && (!swept || t.gcn_prescale == 2))
                rc = get_packed_ids(ds, stream, plan.handle, B, std::max(1, std::min(kWave, t.groups_per_chunk * B)), true, true, &ids, &off);
            if (rc != GNNA_OK) return rc;
            if (phases_out && Bs >= 2) phases_out[i] = first_of_width ? Bs : std::max(phases_out[i], Bs);
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

int gnna_forget_plans(const int32_t *column_index)
{
    if (!column_index) return fail(GNNA_ERR_INVALID_ARGUMENT, "gnna_forget_plans: null column_index");
    (void)release_slice_plans(column_index, true);
    return GNNA_OK;
}

int gnna_last_num_phases(void) { return t_last_phases; }
int gnna_last_num_launches(void) { return t_last_launches; }

#pragma GCC visibility pop
}  // extern "C"
