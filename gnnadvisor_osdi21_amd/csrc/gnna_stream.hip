// gnna_stream.hip -- the streaming form of the neighbor-group aggregation kernel and the sliced
// (source-range) schedule built on it.  CDNA4 / gfx950 only.
//
// Same computation as agg_kernel (gnna_agg.hip; reference GNNAdvisor_kernel.cu:186-259, 620-689):
//   out[part2Node[g], :] += sum_{e in [partPtr[g], partPtr[g+1])} X[colidx[e], :]      (x eps / row factor)
// What is different is how a wavefront walks its chunk of G neighbor-groups.
//
// Why: the column-phased schedule of round 1 (one launch per source-id range, per-run cursors) is
// bound by a fixed cost per phase -- measured 70-90 us on the Reddit-like graph even for a phase that
// finds nothing to gather -- because every run costs a chain of dependent round trips (chunk
// descriptors -> cursor -> id tile at the cursor -> rows -> read-modify-write) with nothing else of
// that wavefront in flight, and eight waves per SIMD cannot hide ~12 such trips per chunk.  Four
// phases (14.9 MB slices, 60 % L2 hits) were therefore the optimum, although 3.7 MB slices would be
// L2 resident.  Here the work item is (chunk, slice) and costs a fixed number of round trips
// whatever the number of groups or rows in the chunk:
//
//   1. descriptors: part2Node / partPtr of the G groups and the group's 16 slice counts -- one
//      coalesced load each, all in flight together;
//   2. the pieces of all G groups that fall into this slice are laid out as one list of wave-wide
//      row loads (a piece is padded to a multiple of RPI = rows per load, so that a load never mixes
//      destination rows), the column ids of up to 256 list slots are fetched together and parked
//      in LDS (1 KiB per wavefront);
//   3. the list is streamed: U row loads are always in flight, across piece and row boundaries;
//      partial rows accumulate in VGPRs and are folded and flushed when the list says "last load of
//      this destination row" (scalar bit mask).  Loads are never predicated (a padded slot re-reads
//      slot 0's row and is masked at the add), so the compiler's in-order vmcnt bookkeeping stays
//      exact and the ring really stays full.
//
// Sliced schedule: the source rows are cut into S = 32 equal slices; `cnt[g][f]` (uint8) is the
// number of column ids of group g that lie in slice f, computed once per graph by slice_count_kernel
// and kept in a small library cache.  A slice phase takes, from every group, the id positions
// [sum_{f'<lo} cnt, sum_{f'<hi} cnt) clamped to the group's length (the last phase takes the rest):
// whatever bytes `cnt` holds, the phases partition every group's positions exactly, so a stale cache
// entry (same addresses, different contents) can only cost locality, never correctness -- the same
// rule as for the graph hints.  With sorted ids (the loader's CSR) positions and slices coincide.
// All flushes of the sliced schedule are float atomics on a zero-filled output, so the phases need
// no order: they are ONE launch whose blocks are numbered slice-major, i.e. the chip sweeps the
// slices in time (each XCD's L2 holds about one slice) without launch gaps or per-phase tails, and
// there is no state between library calls (the cursors are gone).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <mutex>
#include <vector>

#include "gnna.h"
#include "gnna_device.h"
#include "gnna_internal.h"

namespace gnna {
namespace {

constexpr int kIdSlots = 512;    // at most this many column ids parked in LDS per wavefront and round
#ifndef GNNA_NARROW_SLOTS
#define GNNA_NARROW_SLOTS 1024
#endif
// (rows of <= 16 floats: 16 rows per load, so 512 slots would be only 32 loads per round; with 1024 a round holds 64
// like every other width -- Reddit-like D = 16: 0.900 -> 0.860 ms, D = 8: 0.857 -> 0.810 ms.  Not for the modes that
// park a second per-slot array in LDS: 3 x 16 KiB per block would cost occupancy)
template <int LPR, int MODE>
constexpr int id_slots() { return (LPR == 4 && MODE != MODE_GCN && MODE != MODE_SDDMM) ? GNNA_NARROW_SLOTS : kIdSlots; }

struct StreamParams {
    const float *X;
    const int32_t *col;
    const int32_t *pp;
    const int32_t *p2n;
    float *Y;
    const uint8_t *cnt;        // [S-1][P] cumulative slice counts, or nullptr: one phase takes everything
    const float *row_scale;    // MODE_GIN: optional per-destination-row factor on top of eps
    const float *deg_row;      // MODE_GCN (per-edge coefficients): degree norm per destination row ...
    const float *deg_col;      // ... and per source row
    const float *A;            // MODE_SDDMM: destination-side features [num_out_rows, D]; Y is then edge_out [nnz]
    const int32_t *flag;       // *flag == seq  <=>  partition is NOT canonical
    int64_t P;
    int64_t num_chunks;
    int64_t blocks_per_phase;  // multiple of kXcds when xcd_remap
    int32_t seq;
    int32_t trust;
    int32_t D;                 // floats per destination row (what is stored)
    int32_t DL;                // floats per gathered row (>= 4: rows narrower than 4 floats are gathered from a staged,
                               // 4-float-strided copy and only their first D floats are stored)
    int32_t ldx;               // row stride of X in floats
    int32_t ldy;               // row stride of Y in floats
    int32_t lda;               // MODE_SDDMM: row stride of A in floats
    int32_t G;
    int32_t S;                 // fine slices of the plan
    int32_t B;                 // phases: phase p covers fine slices [win_lo + p*W/B, win_lo + (p+1)*W/B), W = win_hi - win_lo
    int32_t win_lo, win_hi;    // the fine slices this launch covers: [0, S) unless the call is one of a windowed sequence
                               // (win_hi == S: "and everything that is left")
    int32_t phase_lo;          // first phase of this launch
    int32_t relu;              // 1: rows this work item stores (plain, once) are stored as max(x, 0)
    int32_t plain_ok;          // 1: rows owned by one work item may be written with plain stores
    int32_t xcd_remap;
    // deterministic schedule (gnna_tuning.deterministic): one launch per phase, in order; a row owned by the work item is
    // read-modify-written, the partial of a row shared between chunks goes to det_part[chunk][slot][D] (slot 0: the
    // chunk's first row when it continues from the previous chunk, slot 1: its last row) with det_stamp[chunk][slot] =
    // stamp, and det_fixup_kernel adds the partials of every shared row in chunk order
    float *det_part;
    int32_t *det_stamp;
    int32_t stamp;
    int32_t det;
    float eps;
    // packed column ids of a prepared graph (gnna_prepare_graph): the ids of work item (phase, chunk) contiguous at
    // ids_packed[item_off[phase * num_chunks + chunk] ...], groups in order -- or null: ids are read from `col`
    const int32_t *ids_packed;
    const uint32_t *item_off;
    const int32_t *packed_stale;   // *packed_stale == seq: the prologue found column_index changed since the copy was made
};

// ---- slice counts ---------------------------------------------------------------------------------
// cnt[g][f] = number of column ids of neighbor-group g in source slice f (saturating at 255; the
// consumer clamps, see above).  One wavefront per group tile; also counts the non-empty (group,
// slice) cells for S, S/2, S/4, S/8 slices and the edges, which the launcher uses to pick the
// number of phases.
struct SliceStats {
    unsigned long long cells[kSliceLevels];   // non-empty cells when the S slices are merged in pairs 0 .. 4 times
    unsigned long long edges;
    unsigned long long groups;     // non-empty groups
    unsigned long long span;       // sum over the edges of |column id - destination row|
    unsigned long long near[24];   // near[k]: edges with |column id - destination row| < 256 * 2^(k / 2)
    unsigned long long unsorted;   // neighbor-groups whose column ids are not in non-decreasing order
};

__global__ void __launch_bounds__(kBlock)
slice_count_kernel(const int32_t *__restrict__ col, const int32_t *__restrict__ pp, const int32_t *__restrict__ p2n,
                   int64_t P, uint32_t slice_rows, int S, uint8_t *__restrict__ cnt, SliceStats *stats)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    unsigned long long cells[kSliceLevels] = {0, 0, 0, 0, 0}, edges = 0, groups = 0, span = 0, unsorted = 0;
    unsigned long long near_l = 0;   // lane k < 24 accumulates near[k]
    for (int64_t g = wave; g < P; g += nwaves) {
        const int beg = pp[g], end = pp[g + 1];
        const int row = p2n[g];
        int mine = 0;  // lane f < S accumulates the count of slice f
        uint32_t last_id = 0;          // the previous tile's last id (wave-uniform)
        bool descending = false;
        for (int t = beg; t < end; t += kWave) {
            const bool valid = t + lane < end;
            int f = -1, bucket = 99;
            uint32_t my_id = 0xffffffffu;
            if (valid) {
                const uint32_t id = (uint32_t)__builtin_nontemporal_load(col + t + lane);
                my_id = id;
                f = (int)min(id / slice_rows, (uint32_t)(S - 1));
                const int dist = (int)id - row;
                const unsigned ad = (unsigned)(dist < 0 ? -dist : dist);
                span += (unsigned long long)ad;
                if (ad < 256u) {
                    bucket = 0;
                } else {   // smallest k with ad < 256 * 2^(k/2): half-octave thresholds
                    const int m = 31 - __builtin_clz(ad);
                    bucket = 2 * (m - 8) + (ad < (((1u << m) * 181u) >> 7) ? 1 : 2);
                }
            }
            {
                // sortedness of the group's ids (the windowed entry depends on it): every id against its predecessor
                const uint32_t prev = lane == 0 ? last_id : (uint32_t)__shfl_up((int)my_id, 1);
                if (__ballot(valid && (t + lane > beg) && my_id < prev) != 0) descending = true;
                const int nv = end - t < kWave ? end - t : kWave;
                last_id = (uint32_t)__builtin_amdgcn_readlane((int)my_id, nv - 1);
            }
#pragma unroll
            for (int b = 0; b < 24; b++) {
                const int c = __popcll(__ballot(bucket <= b));
                if (lane == b) near_l += (unsigned long long)c;
            }
            // sorted ids put a tile's edges into few slices: only the slices that occur are counted
            unsigned long long left = __ballot(valid);
            while (left) {
                const int f0 = __builtin_amdgcn_readlane(f, __builtin_ctzll(left));
                const unsigned long long same = __ballot(f == f0);
                if (lane == f0) mine += __popcll(same);
                left &= ~same;
            }
        }
        {
            // cum[f][g] = ids of the group below slice f + 1 = inclusive prefix over lanes 0 .. f
            const int pre = wave_inclusive_scan(lane < S ? mine : 0);
            if (lane < S - 1) cnt[(size_t)lane * (size_t)P + (size_t)g] = (uint8_t)min(pre, 255);
        }
        if (end > beg) {
            // non-empty cells at S, S/2, ... 2 slices (lanes 0 .. S-1 hold the fine counts)
            int v = lane < S ? mine : 0;
#pragma unroll
            for (int lvl = 0; lvl < kSliceLevels; lvl++) {
                const unsigned long long m = __ballot(v > 0 && lane < (S >> lvl));
                cells[lvl] += (unsigned long long)__popcll(m);
                // merge pairs: lane i takes lanes 2i, 2i+1
                const int a = __shfl(v, 2 * lane), b2 = __shfl(v, 2 * lane + 1);
                v = lane < (S >> (lvl + 1)) ? a + b2 : 0;
            }
            edges += (unsigned long long)(end - beg);
            groups += 1;
            if (descending) unsorted += 1;
        }
    }
    // span was accumulated per lane: reduce over the wavefront
    for (int d = 32; d > 0; d >>= 1) span += __shfl_down(span, d);
    if (stats && lane < 24 && near_l) atomicAdd(&stats->near[lane], near_l);
    if (stats && lane == 0) {
        if (span) atomicAdd(&stats->span, span);
        for (int i = 0; i < kSliceLevels; i++)
            if (cells[i]) atomicAdd(&stats->cells[i], cells[i]);
        if (edges) atomicAdd(&stats->edges, edges);
        if (groups) atomicAdd(&stats->groups, groups);
        if (unsorted) atomicAdd(&stats->unsorted, unsorted);
    }
}

// ---- flush ----------------------------------------------------------------------------------------
// park_row: stores a folded row piece (the layout fold_row leaves) into the wavefront's LDS buffer in natural
// float order: float i of the buffer is out[row, d0 + i].  A ragged last piece (shifted back to end at D)
// overlaps its predecessor with identical values.
template <int LPR>
__device__ __forceinline__ void park_row(const typename VecOf<4>::T r, float *__restrict__ buf, int rel_col, bool cvalid,
                                         int lane, int slot)
{
    // (rel_col + k < 0: floats of a shifted piece that belong to the previous dimension sweep, flushed there)
    if constexpr (LPR <= 16) {
        const int at = rel_col + (lane >> 4);
        if ((lane & 15) < LPR && cvalid && at >= 0) buf[at] = r[0];
    } else {
        if (slot == 0 && cvalid) {
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (rel_col + k >= 0) buf[rel_col + k] = r[k];
        }
    }
}

// emit_row: writes / adds a parked row piece of `width` floats to dst[0 : width], 64 consecutive floats per
// instruction (one fully covered 256-byte run per wave-wide store / atomic).  how: 0 non-temporal store, 1 float
// atomic add, 2 plain read-add-write (deterministic schedule: nobody else touches the row), 3 plain store
// (a partial row parked in library scratch).
enum { EMIT_STORE = 0, EMIT_ATOMIC = 1, EMIT_RMW = 2, EMIT_PART = 3 };
template <int LPR>
__device__ __forceinline__ void emit_row(const float *__restrict__ buf, float *__restrict__ dst, int width, int how, int lane,
                                         bool relu)
{
#pragma unroll
    for (int i = 0; i < (4 * LPR + kWave - 1) / kWave; i++) {
        const int idx = i * kWave + lane;
        if (idx < width) {
            // (the fused ReLU epilogue applies where the row is complete when it is written: the plain store of a row one
            // work item owns; rows that are added to are left to the fix-up pass behind the kernel)
            if (how == EMIT_STORE) __builtin_nontemporal_store(relu ? fmaxf(buf[idx], 0.f) : buf[idx], dst + idx);
            else if (how == EMIT_ATOMIC) unsafeAtomicAdd(dst + idx, buf[idx]);
            else if (how == EMIT_RMW) dst[idx] = dst[idx] + buf[idx];
            else dst[idx] = buf[idx];
        }
    }
}

// ---- the streaming kernel ---------------------------------------------------------------------------

#ifndef GNNA_STREAM_BLOCK
#define GNNA_STREAM_BLOCK 256
#endif
constexpr int kSBlock = GNNA_STREAM_BLOCK;        // threads per block of stream_kernel
constexpr int kSWaves = kSBlock / kWave;

template <int LPR, int MODE, int U, bool WIDE>
__global__ void __launch_bounds__(kSBlock)
stream_kernel(const StreamParams p)
{
    typedef typename VecOf<4>::T VT;
    typedef typename VecOf<4>::M MT;
    typedef typename std::conditional<WIDE, uint64_t, uint32_t>::type OffT;
    constexpr int RPI = kWave / LPR;                                   // neighbor rows per wave-wide load
    constexpr int RL = (id_slots<LPR, MODE>() / RPI < kWave) ? id_slots<LPR, MODE>() / RPI : kWave;  // loads per round
    static_assert(RL % U == 0, "a round is a whole number of batches");
    // per wavefront: the round's list slots as row offsets into X (bytes; row index when X > 4 GiB)
    __shared__ uint32_t s_off[kSWaves][RL * RPI];
    // MODE_GCN: the per-edge coefficient round(deg_i * deg_j) of every list slot (reference .cu:355,389)
    __shared__ float s_cf[kSWaves][MODE == MODE_GCN ? RL * RPI : 1];
    // folded rows waiting to be written: the float atomics of the sliced schedule are memory-side round
    // trips that sit in the same in-order vmcnt queue as the row loads, so a flush in the middle of the
    // stream would stall the ring until it retires.  Rows are parked here (2 KiB per wavefront) and
    // emitted in a batch between rounds.
    constexpr int PEND = LPR <= 16 ? 8 : (LPR == 32 ? 4 : 2);          // rows parked per wavefront
    constexpr int PEND_FLOATS = LPR * 4;                               // floats per parked row (one dimension sweep)
    // (MODE_SDDMM parks the round's dot products there instead: one float per list slot)
    constexpr int PEND_TOTAL = MODE == MODE_SDDMM ? RL * RPI : PEND * PEND_FLOATS;
    __shared__ float s_pend[kSWaves][PEND_TOTAL];

    const int lane = threadIdx.x & (kWave - 1);
    const int wib = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int slot = lane / LPR;
    const int c = lane % LPR;
    const int D = p.D;
    const int DL = p.DL;
    const bool canonical = p.trust || (*p.flag != p.seq);
    const char *xbase = reinterpret_cast<const char *>(p.X);
    const uint32_t row_bytes32 = (uint32_t)p.ldx * 4u;
    uint32_t *offs = s_off[wib];
    float *cfs = s_cf[wib];
    float *pend = s_pend[wib];
    int pend_meta = 0;      // lane q: (row << 2 | atomic) of parked row q
    int npend = 0;

    // block -> (phase, chunk): blocks are numbered phase-major, inside a phase every XCD gets one
    // contiguous range of chunks (blocks land on XCD blockIdx % 8)
    const int64_t bpp = p.blocks_per_phase;
    const int phase = p.phase_lo + (int)((int64_t)blockIdx.x / bpp);
    int64_t item = (int64_t)blockIdx.x % bpp;
    if (p.xcd_remap) item = (item % kXcds) * (bpp / kXcds) + item / kXcds;
    const int64_t chunk = item * kSWaves + wib;
    if (chunk >= p.num_chunks) return;
    const int G = p.G;
    const int64_t g0 = chunk * G;
    const int ng = (int)(p.P - g0 < (int64_t)G ? p.P - g0 : (int64_t)G);

    // ---- 1. descriptors (all loads in flight together) ---------------------------------------------
    const bool gl = lane < ng;
    const int my_row = gl ? p.p2n[g0 + lane] : -1;
    const int pa = gl ? p.pp[g0 + lane] : 0;
    const int pb = gl ? p.pp[g0 + lane + 1] : 0;
    // cumulative slice counts of the group at the phase's two slice boundaries (phase-major byte arrays)
    const int W = p.win_hi - p.win_lo;
    const int f_lo = p.cnt ? p.win_lo + phase * W / p.B : 0, f_hi = p.cnt ? p.win_lo + (phase + 1) * W / p.B : 1;
    int cum_lo = 0, cum_hi = 0x7fffffff;
    if (p.cnt && gl) {
        if (f_lo > 0) cum_lo = p.cnt[(size_t)(f_lo - 1) * (size_t)p.P + (size_t)(g0 + lane)];
        if (f_hi < p.S) cum_hi = p.cnt[(size_t)(f_hi - 1) * (size_t)p.P + (size_t)(g0 + lane)];
    }
    int prev_row = -1, next_row = -1;
    if (g0 > 0) prev_row = p.p2n[g0 - 1];
    if (g0 + ng < p.P) next_row = p.p2n[g0 + ng];
    const bool packed = p.ids_packed != nullptr && *p.packed_stale != p.seq;
    const uint32_t item_base = packed ? p.item_off[(size_t)phase * (size_t)p.num_chunks + (size_t)chunk] : 0u;
    const int32_t *__restrict__ ids = packed ? p.ids_packed : p.col;

    const int len = pb > pa ? pb - pa : 0;
    // positions [beg, end) of the group belong to this phase.  The stored bytes are non-decreasing in the
    // slice index (the counting kernel writes prefixes), so consecutive phases meet exactly whatever the
    // bytes are: stale counts cost locality, not correctness.
    const int beg = cum_lo < len ? cum_lo : len;
    int end = cum_hi < len ? cum_hi : len;
    end = end > beg ? end : beg;
    const int n_own = gl ? end - beg : 0;  // edges of this group in this phase
    if (__ballot(n_own > 0) == 0) return;  // nothing of this chunk in this phase

    // destination-row segments and who flushes
    const int up_row = __shfl_up(my_row, 1);
    const bool seg_start = gl && (lane == 0 || my_row != up_row || !canonical);
    const unsigned long long SS = __ballot(seg_start);
    const unsigned long long upto = (2ull << lane) - 1ull;             // lanes <= lane
    const unsigned long long above = ~upto;                             // lanes > lane
    // pieces of consecutive groups of one row that are adjacent in the edge array (sorted ids: the row's
    // part of the slice) are merged into one piece, headed by the first: no padding, no bookkeeping between them
    // (packed ids: the item's edges are contiguous in group order, so consecutive non-empty groups always are)
    const int up_end = __shfl_up(pa + end, 1), up_n = __shfl_up(n_own, 1);
    // (MODE_SDDMM writes edge_out at the edges' ORIGINAL positions: its pieces merge only where those are adjacent too -- which,
    // with sorted ids, is wherever the aggregation's merge)
    const bool cont = n_own > 0 && !seg_start && up_n > 0 && ((packed && MODE != MODE_SDDMM) || up_end == pa + beg);
    const unsigned long long NE = __ballot(n_own > 0 && !cont);        // piece heads
    const int n_cum = wave_inclusive_scan(n_own);
    const int own_beg = packed ? (int)item_base + (n_cum - n_own) : pa + beg;    // first edge of this group's part
    const unsigned long long heads_above = NE & above;
    const int next_head = heads_above ? __builtin_ctzll(heads_above) : 64;
    const int chain_cum = __shfl(n_cum, next_head - 1);   // (every lane takes part: the sources are not heads)
    const int n = (n_own > 0 && !cont) ? chain_cum - n_cum + n_own : 0;
    const unsigned long long ne_above = NE & above;
    const int next_ne = ne_above ? __builtin_ctzll(ne_above) : 64;
    // a segment starts in (lane, next_ne]  <=>  this piece is the last non-empty one of its row segment
    const unsigned long long between = SS & above & (next_ne < 63 ? ((2ull << next_ne) - 1ull) : ~0ull);
    const bool last_in_seg = n > 0 && (next_ne == 64 || between != 0);
    const int seg_first = 63 - __builtin_clzll((SS & upto) | 1ull);
    const unsigned long long ss_above = SS & above;
    const int seg_last = ss_above ? __builtin_ctzll(ss_above) - 1 : ng - 1;
    const bool shared = canonical && ((seg_first == 0 && prev_row == my_row) || (seg_last == ng - 1 && next_row == my_row));
    const int use_atomic_l = (shared || !canonical || (!p.plain_ok && !p.det)) ? 1 : 0;

    // non-empty pieces compacted to lanes 0 .. R-1 (forward permute to the piece's rank); piece r then
    // holds its first edge, edge count, row + flags, and the inclusive prefix of its wave-wide loads
    const int rank = __popcll(NE & (upto >> 1));
    const int R = __popcll(NE);
    const int dst = (n > 0 ? rank : 63) << 2;       // empty pieces all land on lane 63 (unused unless R == 64, then none is empty)
    const int c_pbeg = __builtin_amdgcn_ds_permute(dst, own_beg);
    // MODE_SDDMM with packed ids: where the piece's edges sit in column_index / edge_out (own_beg is its place in the copy)
    const int c_obeg = MODE == MODE_SDDMM ? __builtin_amdgcn_ds_permute(dst, pa + beg) : 0;
    const int t_n = __builtin_amdgcn_ds_permute(dst, n);   // (executed by every lane: the senders are not the receivers)
    const int c_n = lane < R ? t_n : 0;
    const int c_meta = __builtin_amdgcn_ds_permute(dst, (my_row << 2) | (last_in_seg ? 2 : 0) | use_atomic_l);
    const int c_nl = (c_n + RPI - 1) / RPI;
    const int c_offI = wave_inclusive_scan(c_nl);
    const int c_offX = c_offI - c_nl;
    const int L = __builtin_amdgcn_readlane(c_offI, kWave - 1);

    for (int d0 = 0; d0 < DL; d0 += 4 * LPR) {
        // lane c owns the 4 floats starting at dcol; a ragged last piece is shifted back to end at DL
        const int piece = d0 + c * 4;
        const bool cvalid = piece < DL;
        int dcol = piece, shift = 0;
        if (piece + 4 > DL && cvalid) { dcol = DL - 4; shift = piece - dcol; }
        (void)shift;
        const uint32_t col_off = (uint32_t)(cvalid ? dcol : (d0 + 4 <= DL ? d0 : DL - 4)) * 4u;
        VT acc = vzero<4>();
        const int sweep_width = D - d0 < 4 * LPR ? D - d0 : 4 * LPR;      // (D < DL only for rows narrower than 4 floats)
        auto drain = [&]() {
            for (int q = 0; q < npend; q++) {
                const int meta = __builtin_amdgcn_readlane(pend_meta, q);
                const int64_t row = meta >> 2;
                float *dst = p.Y + (size_t)row * (size_t)p.ldy + d0;
                int how = (meta & 1) ? EMIT_ATOMIC : EMIT_STORE;
                if (p.det && canonical) {
                    how = EMIT_RMW;
                    if (meta & 1) {       // shared with a neighbouring chunk: park the partial, det_fixup_kernel adds it in order
                        const int first_row = __builtin_amdgcn_readfirstlane(my_row);
                        const int slot_p = (row == first_row && prev_row == first_row) ? 0 : 1;
                        dst = p.det_part + ((size_t)chunk * 2 + slot_p) * (size_t)D + d0;
                        how = EMIT_PART;
                        if (lane == 0) p.det_stamp[chunk * 2 + slot_p] = p.stamp;
                    }
                }
                emit_row<LPR>(pend + q * PEND_FLOATS, dst, sweep_width, how, lane, p.relu != 0);
            }
            npend = 0;
        };

        for (int r0 = 0; r0 < L; r0 += RL) {
            // ---- 2. this round's loads: lane j describes load r0 + j --------------------------------
            // piece of load J = number of pieces whose inclusive prefix is <= J.  Prefixes of the compacted
            // pieces are strictly increasing: those below r0 are counted with a ballot, those inside the
            // round's window set one bit each in a scalar mask (a short scalar loop), the rest is a popcount.
            const unsigned long long below = __ballot(lane < R && c_offI <= r0);
            unsigned long long inwin = __ballot(lane < R && c_offI > r0 && c_offI <= r0 + RL - 1);
            unsigned long long E = 0;
            while (inwin) {
                const int kk = __builtin_ctzll(inwin);
                inwin &= inwin - 1;
                E |= 1ull << (__builtin_amdgcn_readlane(c_offI, kk) - r0);
            }
            const int k = __popcll(below) + __popcll(E & upto);
            const int k_offX = __shfl(c_offX, k), k_pbeg = __shfl(c_pbeg, k), k_n = __shfl(c_n, k);
            const int k_meta = __shfl(c_meta, k);
            const int J = r0 + lane;
            const bool active = J < L && lane < RL;
            const int i = J - k_offX;
            const int e_j = k_pbeg + i * RPI;
            int o_j = e_j;                                   // MODE_SDDMM: position of the load's first edge in edge_out
            if constexpr (MODE == MODE_SDDMM) o_j = __shfl(c_obeg, k) + i * RPI;
            int v_j = active ? k_n - i * RPI : 0;
            const bool fl_j = active && (k_meta & 2) && v_j <= RPI;   // last load of the last piece of its row
            v_j = v_j > RPI ? RPI : v_j;
            const int row_j = active ? (k_meta >> 2) : 0;     // MODE_SDDMM: destination row of lane j's load
            const unsigned long long FL = __ballot(fl_j);
            const unsigned long long TM = __ballot(active && v_j < RPI);   // loads with padded slots
            const int nr = (L - r0) < RL ? (L - r0) : RL;

            // lane j fetches the RPI column ids of its load (one vector load when the load is full) and parks
            // them in LDS as row offsets; a padded slot repeats slot 0, an unused load reads row 0.  (Plain loads: a
            // non-temporal load costs several times a normal one on this chip -- nt id loads were 3-6 % of the kernel --
            // and the id lines other phases come back for now stay in the L2.)
            if (lane < RL) {
                uint32_t o[RPI];
                // (plain loads up to 128-float rows; rows of 129-256 floats -- one id per wave-wide load, 1 KiB of L2
                // per row -- do better when the ids leave the L2 first: D = 256: 7.13 ms with nt ids, 7.44 plain)
                constexpr bool NT = LPR >= 64;
                auto id_at = [&](int e) -> uint32_t {
                    if constexpr (NT) return (uint32_t)__builtin_nontemporal_load(ids + e);
                    else return (uint32_t)ids[e];
                };
                if (v_j == RPI) {
                    if constexpr (RPI >= 4) {
                        typedef int i32x4 __attribute__((ext_vector_type(4)));
                        typedef i32x4 i32x4u __attribute__((aligned(4)));
#pragma unroll
                        for (int s4 = 0; s4 < RPI; s4 += 4) {
                            const i32x4 t = *reinterpret_cast<const i32x4u *>(ids + e_j + s4);
                            o[s4] = (uint32_t)t[0]; o[s4 + 1] = (uint32_t)t[1]; o[s4 + 2] = (uint32_t)t[2]; o[s4 + 3] = (uint32_t)t[3];
                        }
                    } else {
#pragma unroll
                        for (int s = 0; s < RPI; s++) o[s] = id_at(e_j + s);
                    }
                } else {
                    const uint32_t first = v_j > 0 ? id_at(e_j) : 0u;
#pragma unroll
                    for (int s = 0; s < RPI; s++) {
                        o[s] = first;
                        if (s > 0 && s < v_j) o[s] = id_at(e_j + s);
                    }
                }
                if constexpr (MODE == MODE_GCN) {
                    const float row_deg = v_j > 0 ? p.deg_row[k_meta >> 2] : 0.f;
#pragma unroll
                    for (int s = 0; s < RPI; s++) cfs[lane * RPI + s] = s < v_j ? row_deg * p.deg_col[o[s]] : 0.f;
                }
#pragma unroll
                for (int s = 0; s < RPI; s++) {
                    if constexpr (!WIDE) o[s] *= row_bytes32;
                    offs[lane * RPI + s] = o[s];
                }
            }

            // ---- 3. stream: U unpredicated row loads always in flight ---------------------------------
            auto row_ptr = [&](uint32_t o) -> const MT * {
                if constexpr (WIDE) return reinterpret_cast<const MT *>(xbase + ((uint64_t)o * (uint64_t)row_bytes32 + col_off));
                else return reinterpret_cast<const MT *>(xbase + (o + col_off));
            };
            const int nb = (nr + U - 1) / U;
            VT v[U];
            if constexpr (MODE == MODE_SDDMM) {
                // edge_out[e] = < A[row(e), :], X[col(e), :] >: every ring slot carries the destination row's piece of its
                // load, no accumulation, no flush -- every edge is written once, in the one phase that owns it.
                // Rows of up to 64 floats (LPR <= 16): the wavefront fetches the destination row ONCE per load, one float per
                // lane (256 bytes through the texture path instead of the kilobyte the RPI slots' identical 16-byte reads
                // cost), and every lane picks its four floats up with ds_bpermute.  Wider rows: the slots (two, or one)
                // read their piece directly.
                // (lanes past the row end read inside the row instead of running past A's last row: their value is
                // zeroed in dot_of, the address must still be inside the tensor)
                constexpr bool SHARED_A = LPR <= 16;
                typedef typename std::conditional<SHARED_A, float, VT>::type AT;
                const int a_el = d0 + (lane & (4 * LPR - 1));
                const float *abase = SHARED_A ? p.A + (a_el < D ? a_el : D - 1)
                                              : p.A + (cvalid ? dcol : (d0 + 4 <= DL ? d0 : DL - 4));
                auto a_load = [&](int j) -> AT {
                    const float *ap = abase + (size_t)__builtin_amdgcn_readlane(row_j, j) * (size_t)p.lda;
                    if constexpr (SHARED_A) return *ap;
                    else return *reinterpret_cast<const MT *>(ap);
                };
                const int a_src = (cvalid ? dcol - d0 : 0) << 2;      // (byte index of the lane that holds float dcol)
                // Round 5: consecutive loads of one destination row share its piece.  The loads of a row segment are adjacent
                // in the list (the pieces of a row's groups are merged), so the row is fetched -- and, for rows of <= 64 floats,
                // permuted into place -- at the segment's FIRST load only (and at the first load of a round); `fresh` is a
                // scalar comparison of two lanes' rows, the branch is wave-uniform.  On the Reddit-like graph a segment is
                // ~8 loads: seven of eight fetches and permute groups are gone.
                auto fresh = [&](int j) -> bool {
                    return j == 0 || __builtin_amdgcn_readlane(row_j, j) != __builtin_amdgcn_readlane(row_j, j - 1);
                };
                AT a[U];
                VT av_keep = vzero<4>();
#pragma unroll
                for (int u = 0; u < U; u++) {
                    if (fresh(u)) a[u] = a_load(u);
                    v[u] = *row_ptr(offs[u * RPI + slot]);
                }
                auto dot_of = [&](int u, int j) {
                    if (fresh(j)) {
                        VT av;
                        if constexpr (SHARED_A) {
#pragma unroll
                            for (int k = 0; k < 4; k++)
                                av[k] = __int_as_float(__builtin_amdgcn_ds_bpermute(a_src + 4 * k, __float_as_int(a[u])));
                        } else {
                            av = a[u];
                        }
                        // components that overlap the previous piece (ragged D) and lanes past the row end do not count
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            if (!cvalid || k < shift) av[k] = 0.f;
                        av_keep = av;
                    }
                    const VT prod = v[u] * av_keep;
                    float dot = lane_group_sum<LPR>((prod[0] + prod[1]) + (prod[2] + prod[3]));
                    // parked in LDS (the round's list slot of the edge) and written out after the round: a
                    // 16-byte store per load in the middle of the stream would sit in the ring's vmcnt queue
                    if (c == 0) pend[j * RPI + slot] = dot;
                };
#pragma unroll 1
                for (int b = 0; b + 1 < nb; b++) {
                    const int jn = (b + 1) * U;
                    uint32_t nn[U];
#pragma unroll
                    for (int u = 0; u < U; u++) nn[u] = offs[(jn + u) * RPI + slot];
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        dot_of(u, b * U + u);
                        if (fresh(jn + u)) a[u] = a_load(jn + u);
                        v[u] = *row_ptr(nn[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < U; u++) {      // the last batch only consumes
                    const int j = (nb - 1) * U + u;
                    if (j < nr) dot_of(u, j);
                }
                // (one 16-byte store per load from the lane that described it measured slower than this pass: 1.02 against
                // 0.91 ms at D = 16, 1.92 against 1.90 at D = 64)
#pragma unroll
                for (int q = 0; q < (RL * RPI + kWave - 1) / kWave; q++) {
                    const int f = q * kWave + lane;
                    const int j = f / RPI, sl = f % RPI;
                    const int ej = __shfl(o_j, j), vj = __shfl(v_j, j);
                    if (f < RL * RPI && j < nr && sl < vj) {
                        float dot = pend[f];
                        float *dst = p.Y + ej + sl;
                        if (d0 > 0) dot += *dst;   // wider than one lane sweep: add to the earlier sweeps' part
                        *dst = dot;
                    }
                }
                continue;
            }
#pragma unroll
            for (int u = 0; u < U; u++) v[u] = *row_ptr(offs[u * RPI + slot]);
            auto consume = [&](int u, int j) {
                if ((TM >> j) & 1ull) {
                    const int vj = __builtin_amdgcn_readlane(v_j, j);
                    if (slot >= vj) v[u] = vzero<4>();
                }
                if constexpr (MODE == MODE_GCN) {
                    // the reference rounds coef * x and the accumulation separately (__fmaf_rn(c, x, 0) then +=,
                    // .cu:405); this file is built with -ffp-contract=off
                    const VT tmp = v[u] * cfs[j * RPI + slot];
                    acc += tmp;
                } else {
                    acc += v[u];
                }
                if ((FL >> j) & 1ull) {
                    const int meta = __builtin_amdgcn_readlane(k_meta, j);
                    float scale = p.eps;
                    if constexpr (MODE == MODE_GIN) {
                        if (p.row_scale) scale *= p.row_scale[meta >> 2];
                    }
                    const VT r = fold_row<LPR, MODE>(acc, scale);
                    if (npend == PEND) { drain(); }
                    park_row<LPR>(r, pend + npend * PEND_FLOATS, dcol - d0, cvalid, lane, slot);
                    pend_meta = lane == npend ? meta : pend_meta;
                    npend++;
                    acc = vzero<4>();
                }
            };
            // all batches but the last: consume a load, request the one U positions ahead
#pragma unroll 1
            for (int b = 0; b + 1 < nb; b++) {
                const int jn = (b + 1) * U;
                uint32_t nn[U];
#pragma unroll
                for (int u = 0; u < U; u++) nn[u] = offs[(jn + u) * RPI + slot];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    consume(u, b * U + u);
                    v[u] = *row_ptr(nn[u]);
                }
            }
            // the last batch only consumes (its slots past the round's end hold row 0 and are skipped)
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int j = (nb - 1) * U + u;
                if (j < nr) consume(u, j);
            }
            if (npend > PEND / 2 || r0 + RL >= L) drain();   // between rounds: nothing of the ring waits behind these
        }
    }
}

// ---- deterministic schedule: shared rows ---------------------------------------------------------------
// One wavefront per chunk.  The chunk in which a shared row FIRST appears (its last row continues in the next
// chunk, and did not come from the previous one) is the head of that row's chain: it sums the partials the chunks of
// the chain parked for this phase -- its own (slot 1), then slot 0 of every following chunk that starts with the
// row -- in chunk order, and adds the sum to the output with a plain read-modify-write.  Only partials stamped by
// this phase's launch count (a chunk without edges of the row in this phase wrote none); consumed stamps are cleared.
__global__ void __launch_bounds__(kBlock)
det_fixup_kernel(const int32_t *__restrict__ p2n, int64_t P, int G, int64_t num_chunks, float *__restrict__ Y, int D, int ldy,
                 const float *__restrict__ part, int32_t *__restrict__ stamps, int32_t stamp)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t c = wave; c < num_chunks; c += nwaves) {
        const int64_t g0 = c * G;
        const int64_t g1 = g0 + G < P ? g0 + G : P;
        const int rf = p2n[g0], rl = p2n[g1 - 1];
        const int prev = g0 > 0 ? p2n[g0 - 1] : -1, next = g1 < P ? p2n[g1] : -1;
        const bool head = next == rl && !(rl == rf && prev == rf);
        if (!head) continue;
        for (int d0 = 0; d0 < D; d0 += kWave) {
            const int i = d0 + lane;
            float acc = 0.f;
            // own partial: slot 1 (the row is not the front-shared first row of this chunk)
            if (stamps[c * 2 + 1] == stamp && i < D) acc = part[((size_t)c * 2 + 1) * (size_t)D + i];
            for (int64_t j = c + 1; j < num_chunks; j++) {
                const int64_t h0 = j * G;
                const int64_t h1 = h0 + G < P ? h0 + G : P;
                if (p2n[h0] != rl) break;
                if (stamps[j * 2] == stamp && i < D) acc += part[((size_t)j * 2) * (size_t)D + i];
                if (!(p2n[h1 - 1] == rl && h1 < P && p2n[h1] == rl)) break;      // the row ends inside chunk j
            }
            if (i < D) Y[(size_t)rl * (size_t)ldy + i] += acc;
        }
        // clear what was consumed (a stale stamp must never match a later launch)
        if (lane == 0) {
            if (stamps[c * 2 + 1] == stamp) stamps[c * 2 + 1] = 0;
            for (int64_t j = c + 1; j < num_chunks; j++) {
                const int64_t h0 = j * G;
                const int64_t h1 = h0 + G < P ? h0 + G : P;
                if (p2n[h0] != rl) break;
                if (stamps[j * 2] == stamp) stamps[j * 2] = 0;
                if (!(p2n[h1 - 1] == rl && h1 < P && p2n[h1] == rl)) break;
            }
        }
    }
}

// ---- fused ReLU epilogue: the rows the kernel could not finish -------------------------------------------
// stream_kernel applies max(x, 0) where it stores a row it owns (single pass, plain stores).  Behind it:
//   whole == 0: the rows that two or more work items added to -- a row that continues across a chunk boundary
//               (groups g0 - 1 and g0 of one row, g0 a multiple of G) -- are clamped here, one wavefront per boundary
//               (a hub row spanning many chunks is clamped once per boundary: idempotent);
//   whole != 0, or the partition turned out not to be canonical (every row was added atomically): the whole output.
__global__ void __launch_bounds__(kBlock)
relu_fixup_kernel(float *__restrict__ Y, int64_t N, int D, int ldy, const int32_t *__restrict__ p2n, int64_t P, int G,
                  const int32_t *flag, int32_t seq, int whole)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    if (whole || *flag == seq) {
        for (int64_t r = wave; r < N; r += nwaves) {
            float *row = Y + (size_t)r * (size_t)ldy;
            for (int i = lane; i < D; i += kWave) row[i] = fmaxf(row[i], 0.f);
        }
        return;
    }
    const int64_t num_chunks = (P + G - 1) / G;
    for (int64_t c = 1 + wave; c < num_chunks; c += nwaves) {
        const int64_t g0 = c * G;
        const int r = p2n[g0];
        if (p2n[g0 - 1] != r || r < 0 || r >= N) continue;
        float *row = Y + (size_t)r * (size_t)ldy;
        for (int i = lane; i < D; i += kWave) row[i] = fmaxf(row[i], 0.f);
    }
}

// ---- packed column ids of a prepared graph -------------------------------------------------------------
// In the sliced schedule a work item (chunk of G groups, phase) reads, of every group, only the ids of the phase's
// slices: 64 fragments of a few ids each, every one a whole 128-byte line of `column_index` that the other phases
// fetch again (Reddit-like, 8 phases: ~2.7 GB of the 7.9 GB a step moves are id lines fetched ~6 times).  For a graph
// that gnna_prepare_graph has declared immutable the plan keeps, per (B, G) in use, a copy of the ids in the order the
// kernel consumes them -- phase-major, then chunk, then group, then edge -- and the start of every item in it: an
// item then reads its ids once, contiguously.  (Not for automatic plans: a copy of ids cannot follow a
// column_index that is rewritten in place, the cumulative counts can.)
__device__ __forceinline__ int group_part(const int32_t *__restrict__ pp, const uint8_t *__restrict__ cnt, int64_t P, int S, int B,
                                          int phase, int64_t g, int *first)
{
    const int pa = pp[g], pb = pp[g + 1];
    const int f_lo = phase * S / B, f_hi = (phase + 1) * S / B;
    int cum_lo = 0, cum_hi = 0x7fffffff;
    if (f_lo > 0) cum_lo = cnt[(size_t)(f_lo - 1) * (size_t)P + (size_t)g];
    if (f_hi < S) cum_hi = cnt[(size_t)(f_hi - 1) * (size_t)P + (size_t)g];
    const int len = pb > pa ? pb - pa : 0;
    const int beg = cum_lo < len ? cum_lo : len;
    int end = cum_hi < len ? cum_hi : len;
    end = end > beg ? end : beg;
    *first = pa + beg;
    return end - beg;
}

// counts[phase * num_chunks + chunk] = edges of the item; one wavefront per item
__global__ void __launch_bounds__(kBlock)
item_count_kernel(const int32_t *__restrict__ pp, const uint8_t *__restrict__ cnt, int64_t P, int64_t num_chunks, int G, int S, int B,
                  uint32_t *__restrict__ counts)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t items = num_chunks * B;
    for (int64_t q = wave; q < items; q += nwaves) {
        const int phase = (int)(q / num_chunks);
        const int64_t g = (q % num_chunks) * G + lane;
        int first = 0;
        const int n = (lane < G && g < P) ? group_part(pp, cnt, P, S, B, phase, g, &first) : 0;
        const int tot = __builtin_amdgcn_readlane(wave_inclusive_scan(n), kWave - 1);
        if (lane == 0) counts[q] = (uint32_t)tot;
    }
}

// in-place exclusive prefix of v[0 .. n) plus the total in v[n]; one workgroup
__global__ void __launch_bounds__(1024)
item_scan_kernel(uint32_t *__restrict__ v, int64_t n)
{
    __shared__ uint32_t part[1024];
    const int64_t per = (n + 1023) / 1024;
    const int64_t lo = (int64_t)threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    uint32_t sum = 0;
    for (int64_t i = lo; i < hi; i++) sum += v[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t add = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (int64_t i = lo; i < hi; i++) { const uint32_t c = v[i]; v[i] = run; run += c; }
    if (threadIdx.x == 1023) v[n] = part[1023];
}

// ids_packed[item_off[q] + (edges of the item's earlier groups) + j] = col[first edge of the group's part + j]
__global__ void __launch_bounds__(kBlock)
item_pack_kernel(const int32_t *__restrict__ col, const int32_t *__restrict__ pp, const uint8_t *__restrict__ cnt, int64_t P,
                 int64_t num_chunks, int G, int S, int B, const uint32_t *__restrict__ item_off, int32_t *__restrict__ out)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t items = num_chunks * B;
    for (int64_t q = wave; q < items; q += nwaves) {
        const int phase = (int)(q / num_chunks);
        const int64_t g = (q % num_chunks) * G + lane;
        int first = 0;
        const int n = (lane < G && g < P) ? group_part(pp, cnt, P, S, B, phase, g, &first) : 0;
        const int incl = wave_inclusive_scan(n);
        int32_t *dst = out + item_off[q] + (uint32_t)(incl - n);
        for (int j = 0; j < n; j++) dst[j] = col[first + j];
    }
}

__global__ void __launch_bounds__(kWave)
ids_checksum_kernel(const int32_t *__restrict__ col, int64_t n, const int32_t *__restrict__ pp, int64_t P,
                    unsigned long long *__restrict__ out)
{
    const unsigned long long v = graph_checksum(col, n, pp, P, (int)threadIdx.x);
    if (threadIdx.x == 0) *out = v;
}

// 64-bit hash of ALL of column_index and part_pointers (gnna_tuning.ids_check_every): every entry, tagged with its position,
// goes through a multiply-xorshift mixer and the results are summed -- order-free, so any grid shape and any interleaving of
// the atomics gives the same value.  state[0] = the copy's sample checksum (ids_checksum_kernel), [1] = the hash stored when the
// copy was made, [2] = running sum, [3] = workgroups finished, [4] = 1 once a later hash differed: the copy is never trusted
// again (the prologue's sample check reads [4]).  The last workgroup to finish compares / stores and clears [2], [3].
__device__ __forceinline__ unsigned long long mix_id(unsigned long long pos, uint32_t id, unsigned long long salt)
{
    unsigned long long x = ((pos << 32) | (unsigned long long)id) ^ salt;
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}

__global__ void __launch_bounds__(kBlock)
ids_full_hash_kernel(const int32_t *__restrict__ col, int64_t n, const int32_t *__restrict__ pp, int64_t P,
                     unsigned long long *__restrict__ state, int store, int32_t *stale_flag, int32_t seq)
{
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (int64_t)gridDim.x * blockDim.x;
    unsigned long long sum = 0;
    // the ids four at a time where the array allows it (16-byte loads: this pass is one read of 4 x nnz bytes)
    const bool vec = (reinterpret_cast<uintptr_t>(col) & 15) == 0;
    const int64_t n4 = vec ? n >> 2 : 0;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    for (int64_t i = tid; i < n4; i += nthreads) {
        const i32x4 v = __builtin_nontemporal_load(reinterpret_cast<const i32x4 *>(col) + i);
        const unsigned long long b = (unsigned long long)i << 2;
        sum += mix_id(b, (uint32_t)v.x, 0) + mix_id(b + 1, (uint32_t)v.y, 0) + mix_id(b + 2, (uint32_t)v.z, 0) + mix_id(b + 3, (uint32_t)v.w, 0);
    }
    for (int64_t i = (n4 << 2) + tid; i < n; i += nthreads) sum += mix_id((unsigned long long)i, (uint32_t)col[i], 0);
    for (int64_t i = tid; i <= P; i += nthreads) sum += mix_id((unsigned long long)i, (uint32_t)pp[i], 0x9E3779B97F4A7C15ull);
    for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d);
    __shared__ unsigned long long part[kBlock / kWave];
    __shared__ bool last;
    if ((threadIdx.x & (kWave - 1)) == 0) part[threadIdx.x / kWave] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < kBlock / kWave; w++) t += part[w];
        atomicAdd(&state[2], t);
        __threadfence();
        last = atomicAdd(&state[3], 1ull) == (unsigned long long)gridDim.x - 1ull;
        if (last) {
            __threadfence();
            const unsigned long long total = atomicAdd(&state[2], 0ull);
            if (store) { state[1] = total; state[4] = 0ull; }
            else if (total != state[1]) { state[4] = 1ull; if (stale_flag) *stale_flag = seq; }
            state[2] = 0ull; state[3] = 0ull;
            __threadfence();
        }
    }
}

// The packed copy's per-call sample check for a call that has no prologue to ride in (SDDMM): one wavefront.
__global__ void __launch_bounds__(kWave)
ids_sample_check_kernel(const int32_t *__restrict__ col, int64_t n, const int32_t *__restrict__ pp, int64_t P,
                        const unsigned long long *__restrict__ state, int32_t *stale_flag, int32_t seq)
{
    const unsigned long long now = graph_checksum(col, n, pp, P, (int)threadIdx.x);
    if (threadIdx.x == 0 && (now != state[0] || state[4] != 0ull)) *stale_flag = seq;
}

// ---- plan cache -------------------------------------------------------------------------------------
struct Plan {
    const void *col = nullptr, *pp = nullptr, *p2n = nullptr;
    int64_t P = 0;
    uint32_t slice_rows = 0;
    int device = -1;
    uint8_t *cnt = nullptr;          // (kMaxSlices - 1) * P bytes of cumulative counts, then this plan's SliceStats
    size_t bytes = 0;                // capacity of `cnt` (counts + statistics tail)
    hipEvent_t ready = nullptr;
    hipStream_t made_on = nullptr;
    hipStream_t last_stream = nullptr;
    bool multi_stream = false;       // used from more than one stream: a rebuild in place must wait for the device
    SliceStats stats{};
    bool have_stats = false;
    bool pinned = false;             // made by gnna_prepare_graph: never evicted, only gnna_release_graph drops it
    uint64_t stamp = 0;
    uint32_t uses = 0;
    // packed column ids (pinned plans only), one copy per (phases, groups per chunk) in use
    struct Packed {
        int B = 0, G = 0;
        int32_t *ids = nullptr;          // nnz ids, then (at item_off) B * num_chunks + 1 item starts
        uint32_t *item_off = nullptr;
        unsigned long long *checksum = nullptr;   // 8 words behind the copy: [0] sample checksum of the graph when the copy was made,
                                                  // [1..4] full hash, its accumulator and the "never trust again" mark (ids_full_hash_kernel)
        uint64_t hits = 0;               // launches that found this copy (every ids_check_every-th runs the full hash)
        int64_t num_ids = 0;
        hipEvent_t ready = nullptr;
        hipStream_t made_on = nullptr;
        uint64_t stamp = 0;              // value of the plan's lookup counter at the last use
    };
    std::vector<Packed> packed;
    uint64_t pack_lookups = 0;
    uint64_t pack_no_memory_until = 0;   // lookups below this do not try to build a copy (the last build's hipMalloc failed)
};
constexpr uint64_t kPackedNoMemoryBackoff = 1024;
constexpr int kMaxPacked = 4;   // copies per plan (a GCN / GIN model aggregates at two or three widths)
constexpr uint64_t kPackedKeep = 16;   // a copy looked up within the plan's last 16 lookups is not replaced at a launch (no build storms
                                       // when more phase counts than copies are in use in turn: the extra ones read column_index)
constexpr int kMaxPlans = 32;   // unpinned plans kept (31 bytes per neighbor-group each; Reddit-like: 59 MB): small next to
                                // 288 GB, and a working set of graphs larger than the table would recount on every call
std::vector<Plan *> g_plans;    // pinned plans are unbounded
std::mutex g_plan_mutex;
uint64_t g_plan_clock = 0;
// back-off for partitions that are never seen twice (tensors re-allocated every step, sampled minibatches): after
// kColdStreak plans in a row were evicted unused, automatic plans are not built for the next g_skip misses
constexpr int kColdStreak = 8;
int g_cold_evictions = 0;
int g_skip_builds = 0;
std::atomic<long long> g_counters[CTR_COUNT];

// Device buffers of plans that were dropped while freeing them was not safe -- by gnna_forget_graph (a finalizer may run
// on any thread at any time: hipFree synchronises the device and would invalidate a stream capture in progress), or
// while another thread was between looking a plan up and enqueueing the kernel that reads it.  Freed at the next point
// where the library may synchronise anyway and no aggregation call is in flight.
std::vector<void *> g_dead;
std::atomic<int> g_in_flight{0};

void drain_dead_locked(int allowed_in_flight)
{
    if (g_dead.empty() || g_in_flight.load(std::memory_order_acquire) > allowed_in_flight) return;
    for (void *ptr : g_dead) (void)hipFree(ptr);     // (hipFree waits for the device: kernels still reading it finish first)
    g_dead.clear();
}

size_t stats_offset(int64_t P) { return (((size_t)P * (size_t)(kMaxSlices - 1)) + 255) & ~(size_t)255; }

typedef void (*StreamKernel)(const StreamParams);

template <int LPR, int MODE>
StreamKernel pick_stream_wide(bool wide, int u)
{
    constexpr int RPI = kWave / LPR;
    constexpr int RL = (id_slots<LPR, MODE>() / RPI < kWave) ? id_slots<LPR, MODE>() / RPI : kWave;
    if constexpr (RL % 8 == 0) {
        if (u >= 8) return wide ? stream_kernel<LPR, MODE, 8, true> : stream_kernel<LPR, MODE, 8, false>;
    }
    return wide ? stream_kernel<LPR, MODE, 4, true> : stream_kernel<LPR, MODE, 4, false>;
}

template <int MODE>
StreamKernel pick_stream_lpr(int lpr, bool wide, int u)
{
    switch (lpr) {
    case 4: return pick_stream_wide<4, MODE>(wide, u);
    case 8: return pick_stream_wide<8, MODE>(wide, u);
    case 16: return pick_stream_wide<16, MODE>(wide, u);
    case 32: return pick_stream_wide<32, MODE>(wide, u);
    default: return pick_stream_wide<64, MODE>(wide, u);
    }
}

}  // namespace

void count_event(int which) { g_counters[which].fetch_add(1, std::memory_order_relaxed); }

// Looks up / builds the slice counts of (column_index, part_pointers) for gathers from num_in_rows source
// rows.  A miss runs slice_count_kernel on `stream`; with want_stats the first use also synchronises the stream
// once to read the statistics (never during stream capture: then out->cnt stays null and the caller takes the
// single-pass schedule).  `pin` (gnna_prepare_graph) keeps the plan until gnna_release_graph.
int get_slice_plan(DeviceState *ds, hipStream_t stream, const int32_t *column_index, const int32_t *part_pointers,
                   const int32_t *part2Node, int64_t num_parts, int64_t num_in_rows, bool want_stats, bool pin,
                   SlicePlan *out, uint32_t window_rows)
{
    *out = SlicePlan();
    const int S = kMaxSlices;
    // (window_rows > 0: the fine slices are the caller's source windows -- the windowed entry -- instead of 1/32 of the rows)
    const uint32_t slice_rows = window_rows ? window_rows : slice_rows_for(num_in_rows);
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(g_plan_mutex);
    Plan *hit = nullptr, *victim = nullptr;
    int unpinned = 0;
    for (Plan *pl : g_plans) {
        if (pl->col == column_index && pl->pp == part_pointers && pl->p2n == part2Node && pl->P == num_parts &&
            pl->slice_rows == slice_rows && pl->device == dev) { hit = pl; break; }
        if (!pl->pinned) {
            unpinned++;
            if (pl->device == dev && (!victim || pl->stamp < victim->stamp)) victim = pl;
        }
    }
    bool rebuild = false;
    if (hit && want_stats && !hit->have_stats) rebuild = true;   // built without statistics (forced phase count): count again
    if (!hit || rebuild) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(stream, &cap);
        if (cap != hipStreamCaptureStatusNone) return GNNA_OK;   // no allocation / sync while capturing
        // (never for the windowed entry: its window counts are not an optimisation, the call cannot run without them)
        if (!hit && !pin && want_stats && window_rows == 0 && g_skip_builds > 0) {   // partitions keep changing: do not count this one
            g_skip_builds--;
            count_event(CTR_BACKOFF_SKIPS);
            return GNNA_OK;
        }
        Plan *pl = hit;
        const size_t bytes = stats_offset(num_parts) + sizeof(SliceStats);
        if (!pl) {
            if (unpinned >= kMaxPlans && victim) {
                pl = victim;                                     // least recently used plan of this device
                if (pl->uses <= 1) {
                    if (++g_cold_evictions >= kColdStreak) { g_skip_builds = 64; g_cold_evictions = 0; }
                } else {
                    g_cold_evictions = 0;
                }
            } else {
                pl = new Plan();
                g_plans.push_back(pl);
            }
        }
        // another stream may still read the old counts (or the statistics tail) of a plan that is rebuilt in place
        if (pl->cnt && (pl->multi_stream || (pl->last_stream != stream && pl->uses > 0))) {
            (void)hipDeviceSynchronize();
            count_event(CTR_LAUNCH_SYNCS);
        }
        if (pl->bytes < bytes) {
            drain_dead_locked(1);                            // (about to allocate anyway; this call is the one in flight)
            if (pl->cnt) { (void)hipFree(pl->cnt); count_event(CTR_LAUNCH_FREES); }
            pl->cnt = nullptr; pl->bytes = 0;
            hipError_t e = hipMalloc(reinterpret_cast<void **>(&pl->cnt), bytes);
            count_event(CTR_LAUNCH_MALLOCS);
            if (e != hipSuccess) {
                pl->col = nullptr;
                return fail(GNNA_ERR_HIP, "hipMalloc(slice counts %zu B): %s", bytes, hipGetErrorString(e));
            }
            pl->bytes = bytes;
        }
        if (!pl->ready) (void)hipEventCreateWithFlags(&pl->ready, hipEventDisableTiming);
        // packed ids of the partition this plan described before (pack_ids = 1 on an automatic plan): gone with it
        for (auto &pk : pl->packed) {
            if (pk.ids) { (void)hipFree(pk.ids); count_event(CTR_LAUNCH_FREES); }
            if (pk.ready) (void)hipEventDestroy(pk.ready);
        }
        pl->packed.clear();
        pl->col = column_index; pl->pp = part_pointers; pl->p2n = part2Node; pl->P = num_parts; pl->slice_rows = slice_rows;
        pl->device = dev; pl->have_stats = false; pl->made_on = stream; pl->last_stream = stream; pl->multi_stream = false;
        pl->uses = 0;
        SliceStats *stats_dev = reinterpret_cast<SliceStats *>(pl->cnt + stats_offset(num_parts));
        (void)hipMemsetAsync(stats_dev, 0, sizeof(SliceStats), stream);
        int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((num_parts + kWavesPerBlock - 1) / kWavesPerBlock,
                                                                  (int64_t)ds->num_cus * 16));
        hipLaunchKernelGGL(slice_count_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, stream, column_index,
                           part_pointers, part2Node, num_parts, slice_rows, S, pl->cnt, stats_dev);
        count_event(CTR_PLAN_BUILDS);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { pl->col = nullptr; return fail(GNNA_ERR_HIP, "slice count launch: %s", hipGetErrorString(e)); }
        if (want_stats) {
            e = hipMemcpyAsync(&pl->stats, stats_dev, sizeof(SliceStats), hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            if (!pin) count_event(CTR_LAUNCH_SYNCS);
            if (e != hipSuccess) { pl->col = nullptr; return fail(GNNA_ERR_HIP, "slice statistics: %s", hipGetErrorString(e)); }
            pl->have_stats = true;
        }
        (void)hipEventRecord(pl->ready, stream);
        hit = pl;
    } else if (hit->made_on != stream) {
        // built on another stream: order after it (not from inside a capture: the counting pass was enqueued before the
        // capture began -- a capture needs its warm-up anyway -- and an outside event must not leak into the graph)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(stream, &cap);
        if (cap == hipStreamCaptureStatusNone) (void)hipStreamWaitEvent(stream, hit->ready, 0);
    }
    if (hit->uses > 0 && hit->last_stream != stream) hit->multi_stream = true;
    hit->last_stream = stream;
    hit->uses++;
    if (hit->uses > 1) { g_cold_evictions = 0; g_skip_builds = 0; }   // partitions do come back: count again
    if (pin) hit->pinned = true;
    hit->stamp = ++g_plan_clock;
    out->cnt = hit->cnt;
    out->handle = hit;
    out->pinned = hit->pinned;
    out->S = S;
    out->slice_rows = slice_rows;
    out->stats.valid = hit->have_stats;
    for (int i = 0; i < kSliceLevels; i++) out->stats.cells[i] = (double)hit->stats.cells[i];
    out->stats.edges = (double)hit->stats.edges;
    out->stats.groups = (double)hit->stats.groups;
    out->stats.span = (double)hit->stats.span;
    out->stats.unsorted = (double)hit->stats.unsorted;
    for (int i = 0; i < 24; i++) out->stats.near[i] = (double)hit->stats.near[i];
    return GNNA_OK;
}

int release_slice_plans(const void *column_index, bool deferred)
{
    std::lock_guard<std::mutex> lock(g_plan_mutex);
    int dropped = 0;
    if (!column_index) { g_cold_evictions = 0; g_skip_builds = 0; }
    for (size_t i = 0; i < g_plans.size();) {
        Plan *pl = g_plans[i];
        if (column_index && pl->col != column_index) { i++; continue; }
        if (pl->cnt) g_dead.push_back(pl->cnt);
        if (pl->ready) (void)hipEventDestroy(pl->ready);
        for (auto &pk : pl->packed) {
            if (pk.ids) g_dead.push_back(pk.ids);
            if (pk.ready) (void)hipEventDestroy(pk.ready);
        }
        delete pl;
        g_plans.erase(g_plans.begin() + (long)i);
        dropped++;
    }
    if (!deferred) drain_dead_locked(0);
    return dropped;
}

void begin_launch()
{
    std::lock_guard<std::mutex> lock(g_plan_mutex);     // (ordered against a drain's check of the counter)
    g_in_flight.fetch_add(1, std::memory_order_acq_rel);
}

void end_launch() { g_in_flight.fetch_sub(1, std::memory_order_acq_rel); }

void drain_dead_buffers()
{
    std::lock_guard<std::mutex> lock(g_plan_mutex);
    drain_dead_locked(0);
}

void drop_slice_plans() { (void)release_slice_plans(nullptr, false); }

int launch_ids_sample_check(hipStream_t stream, const int32_t *col, int64_t n, const int32_t *pp, int64_t P,
                            const unsigned long long *state, int32_t *stale_flag, int32_t seq)
{
    hipLaunchKernelGGL(ids_sample_check_kernel, dim3(1), dim3(kWave), 0, stream, col, n, pp, P, state, stale_flag, seq);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? GNNA_OK : fail(GNNA_ERR_HIP, "packed ids check launch: %s", hipGetErrorString(e));
}

static void launch_full_hash(DeviceState *ds, hipStream_t stream, const Plan *pl, int64_t nnz, unsigned long long *state, int store,
                             int32_t *stale_flag, int32_t seq)
{
    const int64_t work = (nnz + 3) / 4 + pl->P + 1;
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((work + kBlock - 1) / kBlock, (int64_t)ds->num_cus * 8));
    hipLaunchKernelGGL(ids_full_hash_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, stream, static_cast<const int32_t *>(pl->col), nnz,
                       static_cast<const int32_t *>(pl->pp), pl->P, state, store, stale_flag, seq);
}

// Packed ids of a plan for (B phases, G groups per chunk): looked up, or -- with may_build, outside a stream
// capture -- built on `stream` (two small kernels + one pass over column_index; the least recently used copy of a
// plan that already holds kMaxPacked is replaced after a device synchronisation -- at a launch only if it has not been
// used for a while, in gnna_prepare_graph (force) always).  *ids stays null when there is none.
int get_packed_ids(DeviceState *ds, hipStream_t stream, void *plan_handle, int B, int G, bool may_build, bool force,
                   const int32_t **ids, const uint32_t **item_off, const unsigned long long **checksum, int64_t *num_ids,
                   int32_t *stale_flag, int32_t seq, int check_every)
{
    *ids = nullptr; *item_off = nullptr;
    if (checksum) *checksum = nullptr;
    if (num_ids) *num_ids = 0;
    if (!plan_handle || B < 2 || G < 1) return GNNA_OK;
    std::lock_guard<std::mutex> lock(g_plan_mutex);
    Plan *pl = nullptr;
    for (Plan *q : g_plans) if (q == plan_handle) { pl = q; break; }
    if (!pl || !pl->cnt) return GNNA_OK;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(stream, &cap);
    pl->pack_lookups++;
    for (auto &pk : pl->packed) {
        if (pk.B == B && pk.G == G && pk.ids) {
            if (pk.made_on != stream && cap == hipStreamCaptureStatusNone) (void)hipStreamWaitEvent(stream, pk.ready, 0);
            pk.stamp = pl->pack_lookups;
            // every check_every-th launch on this copy: the full hash of the graph behind it, on the same stream ahead of the
            // aggregation (not while capturing: a replayed graph would either always or never pay for it)
            if (stale_flag && check_every > 0 && check_every < (1 << 30) && cap == hipStreamCaptureStatusNone &&
                ++pk.hits % (uint64_t)check_every == 0) {
                launch_full_hash(ds, stream, pl, pk.num_ids, pk.checksum, 0, stale_flag, seq);
                count_event(CTR_FULL_HASHES);
            }
            *ids = pk.ids; *item_off = pk.item_off;
            if (checksum) *checksum = pk.checksum;
            if (num_ids) *num_ids = pk.num_ids;
            return GNNA_OK;
        }
    }
    if (!may_build || cap != hipStreamCaptureStatusNone) return GNNA_OK;
    if (!force && pl->pack_lookups < pl->pack_no_memory_until) return GNNA_OK;   // a recent build ran out of memory
    const int64_t num_chunks = (pl->P + G - 1) / G;
    const int64_t items = num_chunks * B;
    // nnz: from the counting pass's statistics (every prepared plan has them); a plan counted without statistics (forced
    // phase count, pack_ids = 1) reads the last part pointer instead -- one 4-byte copy and a stream synchronisation
    int64_t nnz = 0;
    hipError_t e = hipSuccess;
    if (pl->have_stats) {
        nnz = (int64_t)pl->stats.edges;
    } else {
        int32_t last = 0;
        e = hipMemcpyAsync(&last, static_cast<const int32_t *>(pl->pp) + pl->P, sizeof(int32_t), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        count_event(CTR_LAUNCH_SYNCS);
        if (e != hipSuccess) return fail(GNNA_ERR_HIP, "packed ids (edge count): %s", hipGetErrorString(e));
        nnz = last;
    }
    if (nnz <= 0 || nnz > 0x7fffffffLL) return GNNA_OK;
    Plan::Packed *slot = nullptr;
    if ((int)pl->packed.size() < kMaxPacked) {
        pl->packed.emplace_back();
        slot = &pl->packed.back();
    } else {
        for (auto &pk : pl->packed) if (!slot || !pk.ids || (slot->ids && pk.stamp < slot->stamp)) slot = &pk;
        if (slot->ids) {
            if (!force && pl->pack_lookups - slot->stamp < kPackedKeep) return GNNA_OK;   // every copy is in recent use: none for this one
            (void)hipDeviceSynchronize();           // kernels of any stream may still read the copy that goes
            count_event(CTR_LAUNCH_SYNCS);
            (void)hipFree(slot->ids); count_event(CTR_LAUNCH_FREES);
        }
        slot->ids = nullptr; slot->item_off = nullptr;
    }
    const size_t id_bytes = (((size_t)nnz * sizeof(int32_t)) + 255) & ~(size_t)255;
    const size_t off_bytes = ((((size_t)items + 1) * sizeof(uint32_t)) + 15) & ~(size_t)15;
    const size_t bytes = id_bytes + off_bytes + 64;      // (+ the 8 state words of the checksums)
    e = hipMalloc(reinterpret_cast<void **>(&slot->ids), bytes);
    count_event(CTR_LAUNCH_MALLOCS);
    if (e != hipSuccess) {
        // no memory for the copy: the ids are read from column_index.  The slot does not stay behind as an empty entry (four
        // of them would send every later launch through the replace-the-oldest branch: a device synchronisation and
        // another failing hipMalloc each time), and this plan stops trying for a while
        (void)hipGetLastError();
        slot->ids = nullptr; slot->item_off = nullptr; slot->B = 0; slot->G = 0;
        if (slot->ready) { (void)hipEventDestroy(slot->ready); slot->ready = nullptr; }
        if (slot == &pl->packed.back()) pl->packed.pop_back();
        else slot->stamp = pl->pack_lookups;
        pl->pack_no_memory_until = pl->pack_lookups + kPackedNoMemoryBackoff;
        return GNNA_OK;
    }
    slot->item_off = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(slot->ids) + id_bytes);
    slot->checksum = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(slot->ids) + id_bytes + off_bytes);
    slot->num_ids = nnz;
    slot->B = B; slot->G = G; slot->made_on = stream; slot->stamp = pl->pack_lookups;
    if (!slot->ready) (void)hipEventCreateWithFlags(&slot->ready, hipEventDisableTiming);
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((items + kWavesPerBlock - 1) / kWavesPerBlock, (int64_t)ds->num_cus * 16));
    hipLaunchKernelGGL(item_count_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, stream, static_cast<const int32_t *>(pl->pp),
                       pl->cnt, pl->P, num_chunks, G, kMaxSlices, B, slot->item_off);
    hipLaunchKernelGGL(item_scan_kernel, dim3(1), dim3(1024), 0, stream, slot->item_off, items);
    hipLaunchKernelGGL(item_pack_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, stream, static_cast<const int32_t *>(pl->col),
                       static_cast<const int32_t *>(pl->pp), pl->cnt, pl->P, num_chunks, G, kMaxSlices, B, slot->item_off, slot->ids);
    hipLaunchKernelGGL(ids_checksum_kernel, dim3(1), dim3(kWave), 0, stream, static_cast<const int32_t *>(pl->col), nnz,
                       static_cast<const int32_t *>(pl->pp), pl->P, slot->checksum);
    (void)hipMemsetAsync(slot->checksum + 1, 0, 7 * sizeof(unsigned long long), stream);
    launch_full_hash(ds, stream, pl, nnz, slot->checksum, 1, nullptr, 0);
    slot->hits = 0;
    e = hipGetLastError();
    if (e != hipSuccess) return fail(GNNA_ERR_HIP, "packed ids launch: %s", hipGetErrorString(e));
    (void)hipEventRecord(slot->ready, stream);
    count_event(CTR_PACK_BUILDS);
    *ids = slot->ids; *item_off = slot->item_off;
    if (checksum) *checksum = slot->checksum;
    if (num_ids) *num_ids = slot->num_ids;
    return GNNA_OK;
}

// The part of the ReLU epilogue the aggregation kernel leaves behind (nothing when the call has no epilogue).
static int launch_relu_fixup(const StreamLaunch &a, int G, bool whole, hipStream_t stream)
{
    if (!a.relu || a.mode == MODE_SDDMM) return GNNA_OK;
    // (a partition that turns out not to be canonical makes it the whole-output pass on the device: the grid must be able to)
    const int64_t units = std::max<int64_t>(a.num_out_rows, (a.P + G - 1) / G);
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((units + kWavesPerBlock - 1) / kWavesPerBlock, 256 * 8));
    hipLaunchKernelGGL(relu_fixup_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, stream, a.Y, a.num_out_rows, a.D,
                       a.ldy > 0 ? a.ldy : a.D, a.p2n, a.P, G, a.flag, a.seq, whole ? 1 : 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GNNA_ERR_HIP, "epilogue launch: %s", hipGetErrorString(e));
    return GNNA_OK;
}

int launch_stream_epilogue_only(const StreamLaunch &a, hipStream_t stream)
{
    return launch_relu_fixup(a, 1, /*whole=*/true, stream);
}

int launch_stream(const StreamLaunch &a, hipStream_t stream)
{
    StreamParams p;
    p.X = a.X; p.col = a.col; p.pp = a.pp; p.p2n = a.p2n; p.Y = a.Y; p.cnt = a.cnt; p.row_scale = a.row_scale;
    p.deg_row = a.deg_row; p.deg_col = a.deg_col; p.A = a.A;
    p.flag = a.flag; p.P = a.P; p.seq = a.seq; p.trust = a.trust; p.D = a.D; p.ldx = a.ldx;
    p.DL = std::max(a.D, 4); p.ldy = a.ldy > 0 ? a.ldy : a.D; p.lda = a.lda > 0 ? a.lda : a.D;
    p.G = std::max(1, std::min(a.G, kWave));
    p.num_chunks = (a.P + p.G - 1) / p.G;
    int64_t items = (p.num_chunks + kSWaves - 1) / kSWaves;
    p.xcd_remap = a.xcd_remap ? 1 : 0;
    if (p.xcd_remap) items = (items + kXcds - 1) / kXcds * kXcds;
    p.blocks_per_phase = items;
    p.S = a.cnt ? a.S : 1; p.B = a.cnt ? a.B : 1; p.phase_lo = 0;
    p.win_lo = 0; p.win_hi = p.S;
    if (a.cnt && a.win_hi > a.win_lo) { p.win_lo = a.win_lo; p.win_hi = std::min(a.win_hi, p.S); }
    p.plain_ok = a.plain_ok ? 1 : 0;
    // fused ReLU: in the kernel where a row is stored once (single pass over a canonical partition, plain stores), by
    // the fix-up pass for everything else
    const bool relu_in_kernel = a.relu && a.mode != MODE_SDDMM && p.B == 1 && a.plain_ok && !a.det;
    p.relu = relu_in_kernel ? 1 : 0;
    p.eps = a.eps;
    const int64_t grid = items * (int64_t)p.B;
    if (grid > 0x7fffffffLL) return fail(GNNA_ERR_UNSUPPORTED, "aggregation grid too large (%lld blocks)", (long long)grid);
    int lpr = 4;
    const int pieces = (p.DL + 3) / 4;
    while (lpr < 64 && lpr < pieces) lpr <<= 1;
    StreamKernel k = a.mode == MODE_GIN ? pick_stream_lpr<MODE_GIN>(lpr, a.wide, a.U)
                     : (a.mode == MODE_GCN ? pick_stream_lpr<MODE_GCN>(lpr, a.wide, a.U)
                     : (a.mode == MODE_SDDMM ? pick_stream_lpr<MODE_SDDMM>(lpr, a.wide, a.U) : pick_stream_lpr<MODE_SAG>(lpr, a.wide, a.U)));
    p.det = 0; p.det_part = nullptr; p.det_stamp = nullptr; p.stamp = 0;
    p.ids_packed = !a.packed_stale ? nullptr : a.ids_packed; p.item_off = a.item_off;
    p.packed_stale = a.packed_stale;
    if (a.det && a.mode != MODE_SDDMM) {
        // deterministic schedule: the phases are separate launches in order, each followed by the ordered sum of
        // the rows that chunks share
        p.det = 1; p.det_part = a.det_part; p.det_stamp = a.det_stamp;
        int64_t fblocks = std::max<int64_t>(1, std::min<int64_t>((p.num_chunks + kWavesPerBlock - 1) / kWavesPerBlock, 65535));
        for (int ph = 0; ph < p.B; ph++) {
            p.phase_lo = ph;
            p.stamp = (int32_t)((((uint32_t)a.seq & 0x1ffffffu) << 6) | (uint32_t)ph);
            hipLaunchKernelGGL(k, dim3((unsigned)items), dim3(kSBlock), 0, stream, p);
            hipLaunchKernelGGL(det_fixup_kernel, dim3((unsigned)fblocks), dim3(kBlock), 0, stream, a.p2n, a.P, p.G, p.num_chunks,
                               a.Y, a.D, p.ldy, a.det_part, a.det_stamp, p.stamp);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(GNNA_ERR_HIP, "aggregation launch: %s", hipGetErrorString(e));
        return launch_relu_fixup(a, p.G, /*whole=*/true, stream);
    }
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(kSBlock), 0, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GNNA_ERR_HIP, "aggregation launch: %s", hipGetErrorString(e));
    return launch_relu_fixup(a, p.G, /*whole=*/!relu_in_kernel, stream);
}

}  // namespace gnna

extern "C" {
#pragma GCC visibility push(default)
void gnna_runtime_counters(int64_t out[8])
{
    for (int i = 0; i < 8; i++) out[i] = i < gnna::CTR_COUNT ? (int64_t)gnna::g_counters[i].load() : 0;
}
int gnna_runtime_counters_ex(int64_t *out, int capacity)
{
    for (int i = 0; i < capacity && i < gnna::CTR_COUNT; i++) out[i] = (int64_t)gnna::g_counters[i].load();
    return gnna::CTR_COUNT;
}
#pragma GCC visibility pop
}
