// push_probe3.hip -- third version (push_probe2 + a two-stage software pipeline in the consumer: while the adds of group g issue, the
// LDS row reads of group g + 1 and the entry read of group g + 2 are in flight).  push_probe2.hip -- second version of the source-major probe (see push_probe.hip for the form).  What the first version's
// ablations and counters asked for (profiles/r5/push_form_probe_*): the wavefronts were WAITING half of the time, mostly for
// the scalar-cache loads of the edge entries (-0.5 ms without them), and issued 4.6 scalar instructions per edge.  Here
//   * the entries of block b + 1 travel with its rows: one 16-byte vector load per thread while block b is consumed, parked
//     in LDS behind the same barrier, read back with wave-uniform ds_read_b128 (in order with the row reads: the read of
//     group g + 1 is issued before the rows of group g, so one lgkmcnt wait covers exactly what is needed);
//   * an entry is decoded without scalar arithmetic: the LDS address of its row is v_and_or_b32(entry, 0xff00, 4 * lane),
//     s_set_gpr_idx_idx takes the register from the entry's low byte as it is; only the odd entries of a dword pay one shift.
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef PUSH_R
#define PUSH_R 72
#endif
constexpr int kR = PUSH_R;              // accumulator rows per wavefront: v[128 - R .. 127]
constexpr int kBlockRows = 256;
constexpr int kWaves = 16;
constexpr int kEntBytes = 8192;         // entries of one block (16-bit each), all wavefronts: <= 4096
#if PUSH_R == 72
#define ACC0 "v56"
#define NUM_FREE 56
#define ACC_LOW "v56","v57","v58","v59","v60","v61","v62","v63",
#else
#define ACC0 "v64"
#define NUM_FREE 64
#define ACC_LOW
#endif
#define ACC_CLOBBERS ACC_LOW "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75", \
 "v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95", \
 "v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111", \
 "v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127"

struct PushParams {
    const float *X;
    float *Y;
    const int32_t *src_ids;    // per set: distinct sources ascending, padded to 256; + 2 blocks at the end
    const int32_t *blk_off;    // [S + 1]
    const uint32_t *ent_off;   // [num_blocks * 16 + 1 (+ 32 padding)] first entry (16-bit units, multiple of 8) of (block, wavefront)
    const uint32_t *entries;   // 16-bit entries in pairs: slot << 8 | register
    const int32_t *store_row;
    int ldx;
};

__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_vgpr(NUM_FREE)))
void push3_kernel(const PushParams p)
{
    extern __shared__ float stage[];                    // 2 x 64 KB of rows, then 2 x 8 KB of entries
    uint4 *const ent_lds = reinterpret_cast<uint4 *>(stage + 2 * kBlockRows * 64);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int s = blockIdx.x;
    for (int i = 0; i <= kR; i++) {
        const int ii = __builtin_amdgcn_readfirstlane(i);
        asm volatile("s_set_gpr_idx_on %0, gpr_idx(DST)\n\tv_mov_b32 " ACC0 ", 0\n\ts_set_gpr_idx_off" :: "s"(ii) : ACC_CLOBBERS);
    }
    typedef const __attribute__((address_space(4))) uint32_t *cu32;
    typedef const __attribute__((address_space(4))) int32_t *ci32;
    const ci32 blk_off = (ci32)p.blk_off;
    const cu32 ent_off = (cu32)p.ent_off;
    const int b0 = blk_off[s], nblk = blk_off[s + 1] - b0;
    const int q = lane >> 4, c = lane & 15;
    const int my_slot = 16 * wave + 4 * q;
    float4 r0, r1, r2, r3;
    uint4 ev = make_uint4(0, 0, 0, 0);
    int4 ids = make_int4(0, 0, 0, 0);
    auto fetch_ids = [&](int b) {
        ids = *reinterpret_cast<const int4 *>(p.src_ids + (size_t)(b0 + b) * kBlockRows + my_slot);
    };
    // (block range in entries: [eb, ee); this thread's 16 bytes of it)
    auto fetch_rows = [&](uint32_t eb, uint32_t ee) {
        const float *base = p.X + c * 4;
        r0 = *reinterpret_cast<const float4 *>(base + (size_t)ids.x * p.ldx);
        r1 = *reinterpret_cast<const float4 *>(base + (size_t)ids.y * p.ldx);
        r2 = *reinterpret_cast<const float4 *>(base + (size_t)ids.z * p.ldx);
        r3 = *reinterpret_cast<const float4 *>(base + (size_t)ids.w * p.ldx);
        const uint32_t mine = eb + threadIdx.x * 8;                   // 8 entries = 16 bytes per thread
        if (threadIdx.x < kEntBytes / 16 && mine < ee) ev = *reinterpret_cast<const uint4 *>(p.entries + (mine >> 1));
    };
    auto park = [&](int buf) {
        float *dst = stage + buf * (kBlockRows * 64) + my_slot * 64 + c * 4;
        *reinterpret_cast<float4 *>(dst) = r0;
        *reinterpret_cast<float4 *>(dst + 64) = r1;
        *reinterpret_cast<float4 *>(dst + 128) = r2;
        *reinterpret_cast<float4 *>(dst + 192) = r3;
        if (threadIdx.x < kEntBytes / 16) ent_lds[buf * (kEntBytes / 16) + threadIdx.x] = ev;
    };
    // entry offsets of a block, scalar: [base of the block, this wavefront's begin, its end, end of the block]
    uint32_t cur_b = ent_off[(size_t)b0 * kWaves], cur_lo = ent_off[(size_t)b0 * kWaves + wave],
             cur_hi = ent_off[(size_t)b0 * kWaves + wave + 1], cur_e = ent_off[(size_t)(b0 + 1) * kWaves];
    fetch_ids(0); fetch_rows(cur_b, cur_e); park(0);
    fetch_ids(1);
    __syncthreads();
    const unsigned lane4 = lane * 4;
    unsigned mask = 0xff00u;
    asm volatile("" : "+v"(mask));                      // (kept in a VGPR: VOP3 takes one scalar operand, the entry)
    for (int b = 0; b < nblk; b++) {
        const size_t eo = (size_t)(b0 + b + 1) * kWaves;
        const uint32_t nxt_b = ent_off[eo], nxt_lo = ent_off[eo + wave], nxt_hi = ent_off[eo + wave + 1], nxt_e = ent_off[eo + kWaves];
        fetch_rows(nxt_b, nxt_e);                        // rows and entries of block b + 1
        fetch_ids(b + 2);
        const unsigned rows_base = (unsigned)((b & 1) * (kBlockRows * 256));
        const uint4 *eg = ent_lds + (b & 1) * (kEntBytes / 16) + ((cur_lo - cur_b) >> 3);
        const int groups = (int)((cur_hi - cur_lo) >> 3);
        const unsigned la = lane4 + rows_base;
#define ROW(t, w) { unsigned a_; asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(a_) : "s"(w), "v"(mask), "v"(la)); \
                    t = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(stage) + a_); }
#define DECODE(W, H, e) uint32_t W##0 = __builtin_amdgcn_readfirstlane(e.x), W##1 = __builtin_amdgcn_readfirstlane(e.y), \
                                       W##2 = __builtin_amdgcn_readfirstlane(e.z), W##3 = __builtin_amdgcn_readfirstlane(e.w); \
                        uint32_t H##0 = W##0 >> 16, H##1 = W##1 >> 16, H##2 = W##2 >> 16, H##3 = W##3 >> 16;
#define READS(T, W, H) ROW(T##0, W##0) ROW(T##1, H##0) ROW(T##2, W##1) ROW(T##3, H##1) ROW(T##4, W##2) ROW(T##5, H##2) ROW(T##6, W##3) ROW(T##7, H##3)
#define ADDS(T, W, H) asm volatile("s_set_gpr_idx_on %8, gpr_idx(SRC0,DST)\n\tv_add_f32 " ACC0 ", " ACC0 ", %0\n\t" \
                         "s_set_gpr_idx_idx %9\n\tv_add_f32 " ACC0 ", " ACC0 ", %1\n\t" \
                         "s_set_gpr_idx_idx %10\n\tv_add_f32 " ACC0 ", " ACC0 ", %2\n\t" \
                         "s_set_gpr_idx_idx %11\n\tv_add_f32 " ACC0 ", " ACC0 ", %3\n\t" \
                         "s_set_gpr_idx_idx %12\n\tv_add_f32 " ACC0 ", " ACC0 ", %4\n\t" \
                         "s_set_gpr_idx_idx %13\n\tv_add_f32 " ACC0 ", " ACC0 ", %5\n\t" \
                         "s_set_gpr_idx_idx %14\n\tv_add_f32 " ACC0 ", " ACC0 ", %6\n\t" \
                         "s_set_gpr_idx_idx %15\n\tv_add_f32 " ACC0 ", " ACC0 ", %7\n\ts_set_gpr_idx_off" \
                         :: "v"(T##0), "v"(T##1), "v"(T##2), "v"(T##3), "v"(T##4), "v"(T##5), "v"(T##6), "v"(T##7), \
                            "s"(W##0), "s"(H##0), "s"(W##1), "s"(H##1), "s"(W##2), "s"(H##2), "s"(W##3), "s"(H##3) : ACC_CLOBBERS)
        if (groups > 0) {
            // two groups per trip (ping-pong register sets a / b): the reads of the next group are issued before the adds of
            // the current one; reads past the wavefront's last group hit rows of the buffer for nothing (any slot is inside it)
            uint4 ea = eg[0], eb = eg[1];
            float ta0, ta1, ta2, ta3, ta4, ta5, ta6, ta7, tb0, tb1, tb2, tb3, tb4, tb5, tb6, tb7;
            DECODE(wa, ha, ea)
            READS(ta, wa, ha)
            for (int g = 0; g < groups; g += 2) {
                ea = eg[g + 2];
                DECODE(wb, hb, eb)
                READS(tb, wb, hb)
                ADDS(ta, wa, ha);
                if (g + 1 >= groups) break;
                eb = eg[g + 3];
                {
                    DECODE(wc, hc, ea)
                    READS(ta, wc, hc)
                    ADDS(tb, wb, hb);
                    // (the "a" set of the next trip is the one just decoded)
                    wa0 = wc0; wa1 = wc1; wa2 = wc2; wa3 = wc3; ha0 = hc0; ha1 = hc1; ha2 = hc2; ha3 = hc3;
                }
            }
        }
#undef ROW
#undef DECODE
#undef READS
#undef ADDS
        park((b + 1) & 1);
        cur_b = nxt_b; cur_lo = nxt_lo; cur_hi = nxt_hi; cur_e = nxt_e;
        __syncthreads();
    }
    const int32_t *rows = p.store_row + ((size_t)s * kWaves + wave) * kR;
    for (int i = 0; i < kR; i++) {
        const int ii = __builtin_amdgcn_readfirstlane(i);
        const int row = __builtin_amdgcn_readfirstlane(rows[i]);
        float v;
        asm volatile("s_set_gpr_idx_on %1, gpr_idx(SRC0)\n\tv_mov_b32 %0, " ACC0 "\n\ts_set_gpr_idx_off" : "=v"(v) : "s"(ii) : ACC_CLOBBERS);
        if (row >= 0) p.Y[(size_t)row * 64 + lane] = v;
    }
}

extern "C" __attribute__((visibility("default")))
int push_launch(const float *X, float *Y, const int32_t *src_ids, const int32_t *blk_off, const uint32_t *ent_off,
                const uint32_t *entries, const int32_t *store_row, int ldx, int num_sets)
{
    const int lds = 2 * kBlockRows * 256 + 2 * kEntBytes + 64;      // (+ the one group a wavefront reads past its last)
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void *)push3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr = true;
    }
    PushParams p{X, Y, src_ids, blk_off, ent_off, entries, store_row, ldx};
    hipLaunchKernelGGL(push3_kernel, dim3(num_sets), dim3(1024), lds, 0, p);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

extern "C" __attribute__((visibility("default"))) int push_rows_per_wave(void) { return kR; }
extern "C" __attribute__((visibility("default"))) int push_max_block_entries(void) { return kEntBytes / 2; }
