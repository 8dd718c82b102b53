// gnna_sddmm.hip -- SDDMM over the neighbor-group partition (build-defined extension of libgnna.so).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "gnna.h"
#include "gnna_device.h"
#include "gnna_internal.h"

namespace gnna {
namespace {

// ---- SDDMM over the neighbor-group partition (build-defined extension) -------------------------
// edge_out[e] = < dst_feat[row(e), :], src_feat[colidx[e], :] >  for every edge e of the partition.
// The reference has no SDDMM kernel (SURVEY.md "three things" #1); BASELINE's north star asks
// for one over the same partition, so this follows the aggregation kernel's shape: a wavefront
// owns a chunk of groups, a run keeps the destination row's piece in registers, source rows are
// gathered RPI per wave-wide load, and the LPR lanes of a slot fold their 4-float partial dot
// products with DPP (quad_perm / row_half_mirror / row_mirror) and permlane swaps.
struct SddmmParams {
    const float *A;       // [n_out, D] destination-side features
    const float *B;       // [n_in, D] source-side features
    const int32_t *col;
    const int32_t *pp;
    const int32_t *p2n;
    float *out;           // [nnz]
    int64_t P;
    int64_t num_chunks;
    int32_t D;
    int32_t G;
    // column-phased schedule (as in the aggregation kernel): launch `phase` handles, for every run,
    // the edges from the run's cursor up to the first source id >= phase_hi
    int32_t *cursor;
    int32_t phase;
    int32_t num_phases;
    int32_t phase_hi;
};

template <int LPR, int U, bool WIDE, bool PHASED>
__global__ void __launch_bounds__(kBlock)
sddmm_kernel(const SddmmParams p)
{
    constexpr int VEC = 4;
    constexpr int RPI = kWave / LPR;
    typedef typename VecOf<VEC>::T VT;
    typedef typename VecOf<VEC>::M MT;
    typedef typename std::conditional<WIDE, uint64_t, uint32_t>::type OffT;
    static_assert(U * RPI <= kWave, "a batch must fit one 64-edge id tile");

    const int lane = threadIdx.x & (kWave - 1);
    const int wib = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int slot = lane / LPR;
    const int c = lane % LPR;
    const int D = p.D;
    const int G = p.G;
    const char *bbase = reinterpret_cast<const char *>(p.B);
    const OffT row_bytes = (OffT)D * (OffT)sizeof(float);

    for (int64_t chunk = (int64_t)blockIdx.x * kWavesPerBlock + wib; chunk < p.num_chunks;
         chunk += (int64_t)gridDim.x * kWavesPerBlock) {
        const int64_t g0 = chunk * G;
        const int ng = (int)(p.P - g0 < (int64_t)G ? p.P - g0 : (int64_t)G);
        const int my_row = lane < ng ? p.p2n[g0 + lane] : -1;
        const int my_pp = lane <= ng ? p.pp[g0 + lane] : 0;
        // consecutive groups of one destination row form a run: one load of the row's piece, one
        // contiguous edge segment.  Every edge is written exactly once, so there is no flush to share
        // and any partition (canonical or not) is handled alike.
        const int up_row = __shfl_up(my_row, 1);
        const bool is_start = lane < ng && (lane == 0 || my_row != up_row);
        unsigned long long starts = __ballot(is_start);
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
            int se = __builtin_amdgcn_readlane(my_pp, je);
            if constexpr (PHASED) {
                if (p.phase > 0) sb = __builtin_amdgcn_readlane(my_cur, js);
                if (p.phase + 1 < p.num_phases) {
                    // end of this phase's piece: the contiguous prefix of ids below the bound
                    int pe = sb;
                    while (pe < se) {
                        const int nv = se - pe < kWave ? se - pe : kWave;
                        int id = 0x7fffffff;
                        if (lane < nv) id = __builtin_nontemporal_load(p.col + pe + lane);
                        const unsigned long long below = __ballot(lane < nv && id < p.phase_hi);
                        const int take = below == ~0ull ? kWave : __builtin_ctzll(~below);
                        pe += take;
                        if (take < nv) break;
                    }
                    se = pe;
                }
                if (lane == js) new_cur = se;
                if (sb >= se) continue;
            }
            for (int d0 = 0; d0 < D; d0 += VEC * LPR) {
                const int piece = d0 + c * VEC;
                const bool cvalid = piece < D;
                int dcol = piece, shift = 0;
                if (piece + VEC > D && cvalid) { dcol = D - VEC; shift = piece - dcol; }
                const OffT col_off = (OffT)(cvalid ? dcol : (d0 + VEC <= D ? d0 : D - VEC)) * (OffT)sizeof(float);
                // destination row piece; components that overlap the previous piece (ragged D) and
                // lanes past the row end are zeroed so that they do not contribute to the dot product
                VT a = vzero<VEC>();
                if (cvalid) a = *reinterpret_cast<const MT *>(p.A + (size_t)row * D + dcol);
#pragma unroll
                for (int k = 0; k < VEC; k++)
                    if (k < shift) a[k] = 0.f;
                for (int t = sb; t < se; t += kWave) {
                    const int nv = se - t < kWave ? se - t : kWave;
                    int id = 0;
                    if (lane < nv) id = __builtin_nontemporal_load(p.col + t + lane);
#pragma unroll 1
                    for (int b = 0; b < nv; b += U * RPI) {
                        VT v[U];
                        int nid[U];
#pragma unroll
                        for (int u = 0; u < U; u++) nid[u] = __shfl(id, b + u * RPI + slot);
#pragma unroll
                        for (int u = 0; u < U; u++) {
                            v[u] = vzero<VEC>();
                            if (b + u * RPI + slot < nv)
                                v[u] = *reinterpret_cast<const MT *>(bbase + (OffT)((OffT)nid[u] * row_bytes + col_off));
                        }
#pragma unroll
                        for (int u = 0; u < U; u++) {
                            const VT prod = v[u] * a;
                            float dot = lane_group_sum<LPR>((prod[0] + prod[1]) + (prod[2] + prod[3]));
                            const int el = b + u * RPI + slot;
                            if (c == 0 && el < nv) {
                                float *dst = p.out + t + el;
                                if (d0 > 0) dot += *dst;  // wider than one lane sweep: add to the earlier chunks
                                *dst = dot;
                            }
                        }
                    }
                }
            }
        }
        if constexpr (PHASED) {
            if (is_start && p.phase + 1 < p.num_phases) p.cursor[g0 + lane] = new_cur;
        }
    }
}

typedef void (*SddmmKernel)(const SddmmParams);

template <int LPR>
SddmmKernel pick_sddmm(bool wide, bool phased)
{
    constexpr int U = LPR < 4 ? LPR : 4;
    if (phased) return wide ? sddmm_kernel<LPR, U, true, true> : sddmm_kernel<LPR, U, false, true>;
    return wide ? sddmm_kernel<LPR, U, true, false> : sddmm_kernel<LPR, U, false, false>;
}

int launch_sddmm(const float *dst_feat, const float *src_feat, const int32_t *column_index,
                 const int32_t *part_pointers, const int32_t *part2Node, float *edge_out,
                 int64_t num_out_rows, int64_t num_in_rows, int dim, int64_t num_parts, int partSize,
                 void *stream_v)
{
    if (num_out_rows < 0 || num_in_rows < 0 || dim < 0 || num_parts < 0)
        return fail(GNNA_ERR_INVALID_ARGUMENT, "negative size");
    if (num_out_rows >= ((int64_t)1 << 29))   // (a destination row travels with two flag bits in one 32-bit register)
        return fail(GNNA_ERR_UNSUPPORTED, "%lld destination rows in one call (at most 536870911): shard the rows",
                    (long long)num_out_rows);
    if (num_parts == 0 || dim == 0) return GNNA_OK;
    if (dim < 4) return fail(GNNA_ERR_UNSUPPORTED, "sddmm needs dim >= 4 (got %d)", dim);
    if (!dst_feat || !src_feat || !column_index || !part_pointers || !part2Node || !edge_out)
        return fail(GNNA_ERR_INVALID_ARGUMENT, "null pointer");
    DeviceState *ds = nullptr;
    int rc = get_device_state(&ds);
    if (rc != GNNA_OK) return rc;
    gnna_tuning tune;
    gnna_get_tuning(&tune);
    apply_graph_hints(column_index, 0, &tune);
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    const int pieces = (dim + 3) / 4;
    int lpr = 4;
    while (lpr < 64 && lpr < pieces) lpr <<= 1;
    SddmmParams p;
    p.A = dst_feat; p.B = src_feat; p.col = column_index; p.pp = part_pointers; p.p2n = part2Node;
    p.out = edge_out; p.P = num_parts; p.D = dim;
    p.G = std::max(1, std::min(tune.groups_per_chunk, 63));
    p.num_chunks = (num_parts + p.G - 1) / p.G;
    const size_t b_bytes = (size_t)num_in_rows * (size_t)dim * sizeof(float);
    const bool wide = b_bytes > 0xffffffffull;
    if (tune.stream_kernel != 2) {
        // streaming kernel (gnna_stream.hip, MODE_SDDMM): same work items and sliced schedule as the aggregation; an
        // edge belongs to exactly one slice, so the phases need neither atomics nor a zero-filled output
        int B = 1;
        const uint8_t *cnt = nullptr;
        int S = kMaxSlices;
        SlicePlan plan;
        if (tune.column_phases >= 2 && num_in_rows >= kMaxSlices) {
            B = std::min(tune.column_phases, kMaxSlices);
            rc = get_slice_plan(ds, stream, column_index, part_pointers, part2Node, num_parts, num_in_rows, false, false, &plan);
            if (rc != GNNA_OK) return rc;
        } else if (tune.column_phases == 0 && num_parts >= 1024 && num_in_rows >= 64 && b_bytes >= ((size_t)6 << 20)) {
            rc = get_slice_plan(ds, stream, column_index, part_pointers, part2Node, num_parts, num_in_rows, true, false, &plan);
            if (rc != GNNA_OK) return rc;
            if (plan.cnt && plan.stats.valid)
                B = choose_slices(plan.stats, b_bytes, plan.S, plan.slice_rows, num_out_rows, num_in_rows == num_out_rows,
                                  tune.nonlocal_ids == 1);
        }
        cnt = plan.cnt;
        if (cnt) S = plan.S;
        if (!cnt || B < 2) { B = 1; cnt = nullptr; }
        StreamLaunch a;
        a.mode = MODE_SDDMM; a.X = src_feat; a.A = dst_feat; a.col = column_index; a.pp = part_pointers; a.p2n = part2Node;
        a.Y = edge_out; a.cnt = cnt; a.row_scale = nullptr; a.deg_row = nullptr; a.deg_col = nullptr;
        // the canonical-partition flag only decides between stores and atomics for shared rows; SDDMM shares nothing
        int32_t *flag = nullptr;
        a.seq = next_call_seq(ds, &flag);
        a.flag = flag; a.trust = 1; a.P = num_parts; a.D = dim; a.ldx = dim;
        a.G = std::min(64, std::max(1, tune.groups_per_chunk) * B); a.U = 4; a.S = S; a.B = B;
        a.wide = wide; a.plain_ok = true; a.xcd_remap = tune.xcd_remap != 0; a.eps = 1.f;
        return launch_stream(a, stream);
    }
    // the source-side rows are gathered like the aggregation's, so the same column-phase rule applies,
    // but a phase costs more here (per-edge dot-product fold, cursor scan) and the gain is smaller:
    // measured on the Reddit-like graph (D = 64) 1 / 2 / 4 / 6 phases = 2.91 / 2.66 / 3.03 / 3.36 ms,
    // D = 16 and the low-degree products-like graph lose -- automatic mode uses at most two
    int phases = choose_phases(tune, b_bytes, num_parts, partSize);
    if (tune.column_phases == 0) phases = (phases >= 4) ? 2 : 1;
    p.cursor = nullptr; p.phase = 0; p.num_phases = phases; p.phase_hi = 0x7fffffff;
    if (phases > 1) {
        rc = claim_cursors(ds, stream, column_index, part_pointers, 1, 0, 1);
        if (rc != GNNA_OK) return rc;
        void *ws = nullptr;
        rc = get_workspace(ds, stream, 0, (size_t)num_parts * sizeof(int32_t), &ws);
        if (rc != GNNA_OK) return rc;
        p.cursor = static_cast<int32_t *>(ws);
    }
    SddmmKernel k;
    switch (lpr) {
    case 4: k = pick_sddmm<4>(wide, phases > 1); break;
    case 8: k = pick_sddmm<8>(wide, phases > 1); break;
    case 16: k = pick_sddmm<16>(wide, phases > 1); break;
    case 32: k = pick_sddmm<32>(wide, phases > 1); break;
    default: k = pick_sddmm<64>(wide, phases > 1); break;
    }
    int64_t grid = (p.num_chunks + kWavesPerBlock - 1) / kWavesPerBlock;
    grid = std::max<int64_t>(1, std::min<int64_t>(grid, 0x7fffffff));
    const int64_t width = (num_in_rows + phases - 1) / phases;
    for (int ph = 0; ph < phases; ph++) {
        p.phase = ph;
        p.phase_hi = (int32_t)std::min<int64_t>((int64_t)(ph + 1) * width, 0x7fffffff);
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(kBlock), 0, stream, p);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(GNNA_ERR_HIP, "sddmm launch: %s", hipGetErrorString(e));
    }
    return GNNA_OK;
}

}  // namespace
}  // namespace gnna

using namespace gnna;

extern "C" {
#pragma GCC visibility push(default)

int gnna_sddmm_f32(const float *dst_feat, const float *src_feat, const int32_t *column_index,
                   const int32_t *part_pointers, const int32_t *part2Node, float *edge_out,
                   int64_t num_out_rows, int64_t num_in_rows, int dim, int64_t num_parts, int partSize,
                   void *stream)
{
    if (partSize <= 0) return fail(GNNA_ERR_INVALID_ARGUMENT, "partSize must be positive (got %d)", partSize);
    return launch_sddmm(dst_feat, src_feat, column_index, part_pointers, part2Node, edge_out, num_out_rows,
                        num_in_rows, dim, num_parts, partSize, stream);
}

#pragma GCC visibility pop
}  // extern "C"
