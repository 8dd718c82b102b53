// push_probe4.hip -- fourth version: push_probe2 with HALF the block (128 source rows, 32 KB), 24 accumulator rows per wavefront in
// v[40 .. 63] and 40 VGPRs for the compiler, so that TWO 16-wavefront workgroups share a CU (8 wavefronts per SIMD): one workgroup's
// barrier and LDS latencies are covered by the other's work.  Sets shrink to 384 rows.  push_probe2.hip -- second version of the source-major probe (see push_probe.hip for the form).  What the first version's
// ablations and counters asked for (profiles/r5/push_form_probe_*): the wavefronts were WAITING half of the time, mostly for
// the scalar-cache loads of the edge entries (-0.5 ms without them), and issued 4.6 scalar instructions per edge.  Here
//   * the entries of block b + 1 travel with its rows: one 16-byte vector load per thread while block b is consumed, parked
//     in LDS behind the same barrier, read back with wave-uniform ds_read_b128 (in order with the row reads: the read of
//     group g + 1 is issued before the rows of group g, so one lgkmcnt wait covers exactly what is needed);
//   * an entry is decoded without scalar arithmetic: the LDS address of its row is v_and_or_b32(entry, 0xff00, 4 * lane),
//     s_set_gpr_idx_idx takes the register from the entry's low byte as it is; only the odd entries of a dword pay one shift.
#include <hip/hip_runtime.h>
#include <stdint.h>

constexpr int kR = 24;                  // accumulator rows per wavefront: v[39 .. 62]; v63 = dummy
constexpr int kBlockRows = 128;
constexpr int kWaves = 16;
constexpr int kEntBytes = 4096;         // entries of one block (16-bit each), all wavefronts: <= 2048
#define ACC0 "v39"                      /* rows = indices 0 .. 23 -> v39 .. v62, index 24 = v63 = the dummy the padding entries add into */
#define ACC_CLOBBERS "v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63"

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

__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_vgpr(39)))
void push4_kernel(const PushParams p)
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
    const int my_slot = 8 * wave + 2 * q;
    float4 r0, r1;
    uint4 ev = make_uint4(0, 0, 0, 0);
    int2 ids = make_int2(0, 0);
    auto fetch_ids = [&](int b) {
        ids = *reinterpret_cast<const int2 *>(p.src_ids + (size_t)(b0 + b) * kBlockRows + my_slot);
    };
    // (block range in entries: [eb, ee); this thread's 16 bytes of it)
    auto fetch_rows = [&](uint32_t eb, uint32_t ee) {
        const float *base = p.X + c * 4;
        r0 = *reinterpret_cast<const float4 *>(base + (size_t)ids.x * p.ldx);
        r1 = *reinterpret_cast<const float4 *>(base + (size_t)ids.y * p.ldx);
        const uint32_t mine = eb + threadIdx.x * 8;                   // 8 entries = 16 bytes per thread
        if (threadIdx.x < kEntBytes / 16 && mine < ee) ev = *reinterpret_cast<const uint4 *>(p.entries + (mine >> 1));
    };
    auto park = [&](int buf) {
        float *dst = stage + buf * (kBlockRows * 64) + my_slot * 64 + c * 4;
        *reinterpret_cast<float4 *>(dst) = r0;
        *reinterpret_cast<float4 *>(dst + 64) = r1;
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
        uint4 nx = groups > 0 ? eg[0] : make_uint4(0, 0, 0, 0);
        for (int g = 0; g < groups; g++) {
            const uint32_t w0 = __builtin_amdgcn_readfirstlane(nx.x), w1 = __builtin_amdgcn_readfirstlane(nx.y),
                           w2 = __builtin_amdgcn_readfirstlane(nx.z), w3 = __builtin_amdgcn_readfirstlane(nx.w);
            nx = eg[g + 1];                                            // (one group past the end at most: inside the buffer)
            const uint32_t h0 = w0 >> 16, h1 = w1 >> 16, h2 = w2 >> 16, h3 = w3 >> 16;
            float t0, t1, t2, t3, t4, t5, t6, t7;
            const unsigned la = lane4 + rows_base;
            // LDS byte address of an entry's row: (entry & 0xff00) | (4 * lane + buffer base) in ONE vector instruction (the
            // compiler's own choice is s_and / s_lshr on the scalar unit per entry, which the first version was short of)
#define ROW(t, w) { unsigned a_; asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(a_) : "s"(w), "v"(mask), "v"(la)); \
                    t = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(stage) + a_); }
            ROW(t0, w0) ROW(t1, h0) ROW(t2, w1) ROW(t3, h1) ROW(t4, w2) ROW(t5, h2) ROW(t6, w3) ROW(t7, h3)
#undef ROW
            asm volatile("s_set_gpr_idx_on %8, gpr_idx(SRC0,DST)\n\tv_add_f32 " ACC0 ", " ACC0 ", %0\n\t"
                         "s_set_gpr_idx_idx %9\n\tv_add_f32 " ACC0 ", " ACC0 ", %1\n\t"
                         "s_set_gpr_idx_idx %10\n\tv_add_f32 " ACC0 ", " ACC0 ", %2\n\t"
                         "s_set_gpr_idx_idx %11\n\tv_add_f32 " ACC0 ", " ACC0 ", %3\n\t"
                         "s_set_gpr_idx_idx %12\n\tv_add_f32 " ACC0 ", " ACC0 ", %4\n\t"
                         "s_set_gpr_idx_idx %13\n\tv_add_f32 " ACC0 ", " ACC0 ", %5\n\t"
                         "s_set_gpr_idx_idx %14\n\tv_add_f32 " ACC0 ", " ACC0 ", %6\n\t"
                         "s_set_gpr_idx_idx %15\n\tv_add_f32 " ACC0 ", " ACC0 ", %7\n\ts_set_gpr_idx_off"
                         :: "v"(t0), "v"(t1), "v"(t2), "v"(t3), "v"(t4), "v"(t5), "v"(t6), "v"(t7),
                            "s"(w0), "s"(h0), "s"(w1), "s"(h1), "s"(w2), "s"(h2), "s"(w3), "s"(h3) : ACC_CLOBBERS);
        }
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
    const int lds = 2 * kBlockRows * 256 + 2 * kEntBytes + 16;   // 72 KB: two workgroups per CU      // (+ the one group a wavefront reads past its last)
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void *)push4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr = true;
    }
    PushParams p{X, Y, src_ids, blk_off, ent_off, entries, store_row, ldx};
    hipLaunchKernelGGL(push4_kernel, dim3(num_sets), dim3(1024), lds, 0, p);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

extern "C" __attribute__((visibility("default"))) int push_rows_per_wave(void) { return kR; }
extern "C" __attribute__((visibility("default"))) int push_block_rows(void) { return kBlockRows; }
extern "C" __attribute__((visibility("default"))) int push_max_block_entries(void) { return kEntBytes / 2; }
