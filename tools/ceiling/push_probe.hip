// push_probe.hip -- measurement + correctness probe (round 5, VERDICT r4 task 2): the SOURCE-MAJOR ("push") form of the
// aggregation for 64-float rows.  One 16-wavefront workgroup owns a SET of consecutive destination rows; its accumulators
// live in VGPRs (every wavefront owns up to R rows, one register per row, lane = column; indexed with s_set_gpr_idx), the
// distinct source rows the set references are streamed ONCE, ascending, through LDS in blocks of 256 rows (double
// buffered: block b + 1 travels global -> registers while block b is consumed, then registers -> LDS, one barrier per
// block), and every wavefront applies the block's rows to ITS destination rows from a packed (slot, register) edge list.
// No atomics, no zero-fill, every output row stored once, a fixed summation order.  The packed arrays are built by the
// harness (probe_push.py, torch on the device); this file is only the kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>

// PUSH_ABLATE (timing experiments of this probe only, WRONG results): 1 = rows are not read from LDS (constants added),
// 2 = every add goes to register 0 (no s_set_gpr_idx_idx), 4 = the entries are not re-read (one group reused), 8 = no row
// stream (no global loads, no parking), 16 = no barrier per block
#ifndef PUSH_ABLATE
#define PUSH_ABLATE 0
#endif
#ifndef PUSH_R
#define PUSH_R 72                       // accumulator rows per wavefront: v[128 - R .. 127]
#endif
constexpr int kR = PUSH_R;
constexpr int kBlockRows = 256;         // source rows per staged block
constexpr int kWaves = 16;

#if PUSH_R == 72
#define ACC0 "v56"
#define ACC_CLOBBERS "v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75", \
 "v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95", \
 "v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111", \
 "v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127"
#define NUM_FREE_VGPR 56
#else
#error "PUSH_R"
#endif

struct PushParams {
    const float *X;            // [num_src, ldx] source rows
    float *Y;                  // [num_dst, 64]
    const int32_t *src_ids;    // per set: distinct sources ascending, padded to a multiple of 256 (pad = a valid row id); + 2 blocks at the end
    const int32_t *blk_off;    // [S + 1] first block of every set (block = 256 consecutive entries of src_ids)
    const uint32_t *ent_off;   // [num_blocks * 16 + 1] first entry (in 16-bit units, multiple of 8) of (block, wavefront)
    const uint32_t *entries;   // 16-bit entries in pairs: slot << 8 | register (register kR = dummy)
    const int32_t *store_row;  // [S * 16 * kR] destination row of (set, wavefront, register), -1 = unused
    int ldx;
};

__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_vgpr(NUM_FREE_VGPR)))
void push_kernel(const PushParams p)
{
    extern __shared__ float stage[];                    // 2 x 256 x 64 floats
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int s = blockIdx.x;
    for (int i = 0; i <= kR; i++) {                     // (register kR is the dummy the padding entries add into)
        const int ii = __builtin_amdgcn_readfirstlane(i);
        asm volatile("s_set_gpr_idx_on %0, gpr_idx(DST)\n\tv_mov_b32 " ACC0 ", 0\n\ts_set_gpr_idx_off" :: "s"(ii) : ACC_CLOBBERS);
    }
    // wave-uniform reads (block range, entry offsets, entries) go through the scalar cache: constant address space + a
    // uniform index -> s_load, counted by lgkmcnt -- a vector load here would put an s_waitcnt vmcnt(0) between the
    // prefetch of the next block's rows and their parking, i.e. serialise the row stream with the consumption
    typedef const __attribute__((address_space(4))) uint32_t *cu32;
    typedef const __attribute__((address_space(4))) int32_t *ci32;
    const ci32 blk_off = (ci32)p.blk_off;
    const cu32 ent_off = (cu32)p.ent_off, entries = (cu32)p.entries;
    const int b0 = blk_off[s], nblk = blk_off[s + 1] - b0;
    const int q = lane >> 4, c = lane & 15;
    const int my_slot = 16 * wave + 4 * q;             // this lane loads pieces of slots my_slot .. my_slot + 3
    float4 r0, r1, r2, r3;
    int4 ids = make_int4(0, 0, 0, 0);
    auto fetch_ids = [&](int b) {
        ids = *reinterpret_cast<const int4 *>(p.src_ids + (size_t)(b0 + b) * kBlockRows + my_slot);
    };
    auto fetch_rows = [&]() {
        const float *base = p.X + c * 4;
        r0 = *reinterpret_cast<const float4 *>(base + (size_t)ids.x * p.ldx);
        r1 = *reinterpret_cast<const float4 *>(base + (size_t)ids.y * p.ldx);
        r2 = *reinterpret_cast<const float4 *>(base + (size_t)ids.z * p.ldx);
        r3 = *reinterpret_cast<const float4 *>(base + (size_t)ids.w * p.ldx);
    };
    auto park = [&](int buf) {
        float *dst = stage + buf * (kBlockRows * 64) + my_slot * 64 + c * 4;
        *reinterpret_cast<float4 *>(dst) = r0;
        *reinterpret_cast<float4 *>(dst + 64) = r1;
        *reinterpret_cast<float4 *>(dst + 128) = r2;
        *reinterpret_cast<float4 *>(dst + 192) = r3;
    };
    // (no conditions around the prefetches: src_ids carries two blocks of padding behind the last set, so blocks b + 1 and
    // b + 2 always exist -- past a set's end they are the next set's first blocks, fetched and parked for nothing -- and the
    // straight-line loop body lets the compiler wait for exactly the loads it needs: vmcnt(0) for the ids that arrived a
    // block ago, vmcnt(1) at the parking, nothing in between)
    fetch_ids(0); fetch_rows(); park(0);
    fetch_ids(1);
    __syncthreads();
    for (int b = 0; b < nblk; b++) {
        if (!(PUSH_ABLATE & 8)) {
        fetch_rows();                                   // rows of block b + 1 (their ids arrived during block b - 1)
        fetch_ids(b + 2);
        }
        const float *buf = stage + (b & 1) * (kBlockRows * 64) + lane;
        const size_t eo = (size_t)(b0 + b) * kWaves + wave;
        const uint32_t e_lo = ent_off[eo], e_hi = ent_off[eo + 1];
        // (the entries of the next group of 8 edges are fetched -- scalar cache -- before this group's rows are read from LDS)
        cu32 ep = entries + (e_lo >> 1);
        uint32_t n0 = ep[0], n1 = ep[1], n2 = ep[2], n3 = ep[3];
        for (uint32_t e = e_lo; e < e_hi; e += 8) {
            const uint32_t w0 = n0, w1 = n1, w2 = n2, w3 = n3;
            if (!(PUSH_ABLATE & 4)) {
            ep += 4;
            n0 = ep[0]; n1 = ep[1]; n2 = ep[2]; n3 = ep[3];        // (reads up to 16 bytes past the last group: the array is padded)
            }
#if PUSH_ABLATE & 1
            const float t0 = __int_as_float(w0 + lane), t1 = __int_as_float(w0 * 3 + lane), t2 = __int_as_float(w1 + lane), t3 = __int_as_float(w1 * 3 + lane),
                        t4 = __int_as_float(w2 + lane), t5 = __int_as_float(w2 * 3 + lane), t6 = __int_as_float(w3 + lane), t7 = __int_as_float(w3 * 3 + lane);
#else
            const float t0 = buf[(w0 >> 8 & 0xff) * 64], t1 = buf[(w0 >> 24) * 64];
            const float t2 = buf[(w1 >> 8 & 0xff) * 64], t3 = buf[(w1 >> 24) * 64];
            const float t4 = buf[(w2 >> 8 & 0xff) * 64], t5 = buf[(w2 >> 24) * 64];
            const float t6 = buf[(w3 >> 8 & 0xff) * 64], t7 = buf[(w3 >> 24) * 64];
#endif
#if PUSH_ABLATE & 2
            asm volatile("v_add_f32 " ACC0 ", " ACC0 ", %0\n\tv_add_f32 " ACC0 ", " ACC0 ", %1\n\tv_add_f32 " ACC0 ", " ACC0 ", %2\n\t"
                         "v_add_f32 " ACC0 ", " ACC0 ", %3\n\tv_add_f32 " ACC0 ", " ACC0 ", %4\n\tv_add_f32 " ACC0 ", " ACC0 ", %5\n\t"
                         "v_add_f32 " ACC0 ", " ACC0 ", %6\n\tv_add_f32 " ACC0 ", " ACC0 ", %7"
                         :: "v"(t0), "v"(t1), "v"(t2), "v"(t3), "v"(t4), "v"(t5), "v"(t6), "v"(t7) : ACC_CLOBBERS);
#else
            asm volatile("s_set_gpr_idx_on %8, gpr_idx(SRC0,DST)\n\tv_add_f32 " ACC0 ", " ACC0 ", %0\n\t"
                         "s_set_gpr_idx_idx %9\n\tv_add_f32 " ACC0 ", " ACC0 ", %1\n\t"
                         "s_set_gpr_idx_idx %10\n\tv_add_f32 " ACC0 ", " ACC0 ", %2\n\t"
                         "s_set_gpr_idx_idx %11\n\tv_add_f32 " ACC0 ", " ACC0 ", %3\n\t"
                         "s_set_gpr_idx_idx %12\n\tv_add_f32 " ACC0 ", " ACC0 ", %4\n\t"
                         "s_set_gpr_idx_idx %13\n\tv_add_f32 " ACC0 ", " ACC0 ", %5\n\t"
                         "s_set_gpr_idx_idx %14\n\tv_add_f32 " ACC0 ", " ACC0 ", %6\n\t"
                         "s_set_gpr_idx_idx %15\n\tv_add_f32 " ACC0 ", " ACC0 ", %7\n\ts_set_gpr_idx_off"
                         :: "v"(t0), "v"(t1), "v"(t2), "v"(t3), "v"(t4), "v"(t5), "v"(t6), "v"(t7),
                            "s"(w0 & 0xff), "s"(w0 >> 16 & 0xff), "s"(w1 & 0xff), "s"(w1 >> 16 & 0xff),
                            "s"(w2 & 0xff), "s"(w2 >> 16 & 0xff), "s"(w3 & 0xff), "s"(w3 >> 16 & 0xff) : ACC_CLOBBERS);
#endif
        }
        if (!(PUSH_ABLATE & 8)) park((b + 1) & 1);
        if (!(PUSH_ABLATE & 16)) __syncthreads();
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
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void *)push_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kBlockRows * 256);
        attr = true;
    }
    PushParams p{X, Y, src_ids, blk_off, ent_off, entries, store_row, ldx};
    hipLaunchKernelGGL(push_kernel, dim3(num_sets), dim3(1024), 2 * kBlockRows * 256, 0, p);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

extern "C" __attribute__((visibility("default"))) int push_rows_per_wave(void) { return kR; }
