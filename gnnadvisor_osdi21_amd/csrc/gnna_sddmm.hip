// gnna_sddmm.hip -- SDDMM over the neighbor-group partition (build-defined extension of libgnna.so).
//
//   edge_out[e] = < dst_feat[row(e), :], src_feat[colidx[e], :] >  for every edge e of the partition.
//
// The reference has no SDDMM kernel (SURVEY.md "three things" #1); BASELINE's north star asks for one over the same
// partition.  It runs on the streaming kernel (gnna_stream.hip, MODE_SDDMM): the same work items and sliced schedule as
// the aggregation -- an edge belongs to exactly one slice, so the phases need neither atomics nor a zero-filled output --
// every ring slot carries the destination row's piece of its load, the LPR lanes of a slot fold their 4-float partial
// dot products with DPP (quad_perm / row_half_mirror / row_mirror) and permlane swaps, the round's dot products are
// parked in LDS and written out in coalesced runs.  (Round 1's own chunk-walk SDDMM kernel was retired in 0.4.0 together
// with the chunk-walk aggregation kernel.)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "gnna.h"
#include "gnna_device.h"
#include "gnna_internal.h"

namespace gnna {
namespace {

#ifndef GNNA_SDDMM_FINER
#define GNNA_SDDMM_FINER 1
#endif

int launch_sddmm(const float *dst_feat, int64_t ld_dst, const float *src_feat, int64_t ld_src, const int32_t *column_index,
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
    if (ld_dst < dim || ld_src < dim)
        return fail(GNNA_ERR_INVALID_ARGUMENT, "leading dimensions (%lld, %lld) must be >= dim (%d)", (long long)ld_dst, (long long)ld_src, dim);
    if (ld_dst > 0x7fffffff || ld_src > 0x7fffffff) return fail(GNNA_ERR_UNSUPPORTED, "leading dimension beyond 2^31 - 1 floats");
    if (!dst_feat || !src_feat || !column_index || !part_pointers || !part2Node || !edge_out)
        return fail(GNNA_ERR_INVALID_ARGUMENT, "null pointer");
    DeviceState *ds = nullptr;
    int rc = get_device_state(&ds);
    if (rc != GNNA_OK) return rc;
    LaunchGuard in_flight;
    gnna_tuning tune;
    gnna_get_tuning(&tune);
    apply_graph_hints(column_index, 0, &tune);
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    // the gathered side from the layout the gather likes best (round 6: 1.73 -> 1.62 ms on the Reddit-like graph at D = 64,
    // tools/ceiling/probe_sddmm_floor.py): the staged copy the aggregation makes for hot rows, unless the caller's ld_src is it
    {
        const int64_t est_edges = num_parts * (int64_t)(tune.avg_degree > 0 ? std::min(partSize, tune.avg_degree) : partSize / 2 + 1);
        int ldx = (int)ld_src;
        rc = stage_rows_for_gather(ds, stream, tune, src_feat, ld_src, num_in_rows, dim, est_edges, &src_feat, &ldx);
        if (rc != GNNA_OK) return rc;
        ld_src = ldx;
    }
    const bool wide = (size_t)num_in_rows * (size_t)ld_src * sizeof(float) > 0xffffffffull;
    // what the gather can touch of src_feat (whole 128-byte lines of every row): the size the slicing decision goes by
    const size_t b_bytes = (size_t)num_in_rows * (size_t)std::min<int64_t>(ld_src, (dim * 4 + 127) / 128 * 32) * sizeof(float);
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
        if (plan.cnt && plan.stats.valid) {
            B = choose_slices(plan.stats, b_bytes, plan.S, plan.slice_rows, num_out_rows, num_in_rows == num_out_rows,
                              tune.nonlocal_ids == 1);
            // a (row, slice) piece costs the aggregation a flush of the row; here it costs one more fetch of the destination
            // row's piece and nothing else (every edge is written once, by the phase that owns it) -- slices of half the size
            // pay: Reddit-like D = 64, 8 -> 16 phases 1.62 -> 1.56 ms (tools/ceiling/probe_sddmm_floor.py, profiles/r6/)
            if (B >= 2 && GNNA_SDDMM_FINER) B = std::min(plan.S, 2 * B);
        }
    }
    cnt = plan.cnt;
    if (cnt) S = plan.S;
    if (!cnt || B < 2) { B = 1; cnt = nullptr; }
    StreamLaunch a;
    a.mode = MODE_SDDMM; a.X = src_feat; a.A = dst_feat; a.col = column_index; a.pp = part_pointers; a.p2n = part2Node;
    a.Y = edge_out; a.cnt = cnt; a.row_scale = nullptr; a.deg_row = nullptr; a.deg_col = nullptr;
    // the canonical-partition flag only decides between stores and atomics for shared rows; SDDMM shares nothing
    int32_t *flag = nullptr;
    a.seq = next_call_seq(ds, stream, &flag);
    a.flag = flag; a.trust = 1; a.P = num_parts; a.D = dim; a.ldx = (int)ld_src; a.lda = (int)ld_dst; a.ldy = dim; a.num_out_rows = num_out_rows;
    a.G = std::min(64, std::max(1, tune.groups_per_chunk) * B); a.U = tune.loads_in_flight >= 8 ? 8 : 4; a.S = S; a.B = B;
    a.wide = wide; a.plain_ok = true; a.xcd_remap = tune.xcd_remap != 0; a.eps = 1.f;
    // prepared graph: the ids from the copy the sliced schedule reads contiguously (round 6: the kernel carries the edges'
    // original positions beside their place in the copy, because edge_out is indexed like column_index).  There is no prologue
    // here for the copy's checks to ride in: one wavefront compares the samples first, the full hash runs on its own schedule.
    if (cnt && plan.handle && (tune.pack_ids == 1 || (tune.pack_ids == 0 && plan.pinned))) {
        int32_t *stale_flag = flag + kFlagSlots;
        const unsigned long long *chk = nullptr;
        int64_t chk_n = 0;
        rc = get_packed_ids(ds, stream, plan.handle, B, std::max(1, std::min(a.G, kWave)), true, false, &a.ids_packed, &a.item_off,
                            &chk, &chk_n, stale_flag, a.seq, tune.ids_check_every);
        if (rc != GNNA_OK) return rc;
        if (a.ids_packed) {
            a.packed_stale = stale_flag;
            rc = launch_ids_sample_check(stream, column_index, chk_n, part_pointers, num_parts, chk, stale_flag, a.seq);
            if (rc != GNNA_OK) return rc;
            count_event(CTR_PACKED_LAUNCHES);
        }
    }
    return launch_stream(a, stream);
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
    return launch_sddmm(dst_feat, dim, src_feat, dim, column_index, part_pointers, part2Node, edge_out, num_out_rows,
                        num_in_rows, dim, num_parts, partSize, stream);
}

int gnna_sddmm_ld_f32(const float *dst_feat, int64_t ld_dst, const float *src_feat, int64_t ld_src,
                      const int32_t *column_index, const int32_t *part_pointers, const int32_t *part2Node, float *edge_out,
                      int64_t num_out_rows, int64_t num_in_rows, int dim, int64_t num_parts, int partSize, void *stream)
{
    if (partSize <= 0) return fail(GNNA_ERR_INVALID_ARGUMENT, "partSize must be positive (got %d)", partSize);
    return launch_sddmm(dst_feat, ld_dst, src_feat, ld_src, column_index, part_pointers, part2Node, edge_out, num_out_rows,
                        num_in_rows, dim, num_parts, partSize, stream);
}

#pragma GCC visibility pop
}  // extern "C"
