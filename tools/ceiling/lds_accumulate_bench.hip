// lds_accumulate_bench.hip -- measurement only (round 5, VERDICT r4 task 2): what does ONE "add a 64-float source row
// into a destination row of the set's accumulators" cost per CU, in the forms a source-major (push) aggregation could
// use?  One 1024-thread workgroup per CU (the sweep kernel's geometry), 512 accumulator rows of 64 floats in LDS
// (128 KB); every wavefront holds one source row in a VGPR (lane = column) and applies it to pseudo-random rows.
//   mode 0: ds_add_f32 (no return)                       -- LDS float atomic, correct under any interleaving
//   mode 1: ds_read_b32 + v_add + ds_write_b32            -- plain read-modify-write (needs exclusive rows)
//   mode 2: ds_read_addtid_b32 + v_add + ds_write_addtid  -- the same through M0 (no address VGPR)
//   mode 3: ds_read_b32 only (the staged-source read of the register-accumulator design)
//   mode 4: ds_add_rtn_f32 (returning)                    -- for reference
//   mode 6: 8 x ds_read_b32 of staged source rows in flight, then 8 adds into accumulator rows kept in VGPRs v64..v127,
//           register-indexed through s_set_gpr_idx (the register-accumulator design: no LDS write at all)
// Prints LDS-array cycles per row operation per CU (wall cycles x CUs busy / operations).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kRows = 512;

template <int MODE>
__global__ __launch_bounds__(1024) void bench(const int *__restrict__ rows, int per_wave, float *__restrict__ out)
{
    extern __shared__ float acc[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < kRows * 64; i += 1024) acc[i] = 0.f;
    __syncthreads();
    float v = 1.0f + lane * 1e-3f;
    const int *mine = rows + ((size_t)blockIdx.x * 16 + wave) * per_wave;
    float sink = 0.f;
    for (int i = 0; i < per_wave; i += 8) {
        int r[8];
#pragma unroll
        for (int k = 0; k < 8; k++) r[k] = __builtin_amdgcn_readfirstlane(mine[i + k]);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const unsigned addr = (unsigned)(r[k] * 256 + lane * 4);
            if constexpr (MODE == 0) {
                asm volatile("ds_add_f32 %0, %1" :: "v"(addr), "v"(v) : "memory");
            } else if constexpr (MODE == 1) {
                float t;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_add_f32 %0, %0, %2\n\tds_write_b32 %1, %0"
                             : "=&v"(t) : "v"(addr), "v"(v) : "memory");
            } else if constexpr (MODE == 2) {
                float t;
                unsigned keep;
                asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tds_read_addtid_b32 %0\n\ts_waitcnt lgkmcnt(0)\n\t"
                             "v_add_f32 %0, %0, %3\n\tds_write_addtid_b32 %0\n\ts_mov_b32 m0, %1"
                             : "=&v"(t), "=&s"(keep) : "s"(r[k] * 256), "v"(v) : "memory");
            } else if constexpr (MODE == 3) {
                float t;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(t) : "v"(addr) : "memory");
                sink += t;
            } else {
                float t;
                asm volatile("ds_add_rtn_f32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(t) : "v"(addr), "v"(v) : "memory");
                sink += t;
            }
        }
    }
    __syncthreads();
    float s = sink;
    for (int i = threadIdx.x; i < kRows * 64; i += 1024) s += acc[i];
    if (s == 12345.678f) out[blockIdx.x] = s;
}

// mode 3 without the per-op wait: 8 reads in flight, then 8 adds into registers (how a consumer would really run)
__global__ __launch_bounds__(1024) void bench_read8(const int *__restrict__ rows, int per_wave, float *__restrict__ out)
{
    extern __shared__ float acc[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < kRows * 64; i += 1024) acc[i] = 1.f;
    __syncthreads();
    const int *mine = rows + ((size_t)blockIdx.x * 16 + wave) * per_wave;
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int i = 0; i < per_wave; i += 8) {
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; k++) t[k] = acc[__builtin_amdgcn_readfirstlane(mine[i + k]) * 64 + lane];
        a0 += t[0] + t[4]; a1 += t[1] + t[5]; a2 += t[2] + t[6]; a3 += t[3] + t[7];
    }
    float s = a0 + a1 + a2 + a3;
    if (s == 12345.678f) out[blockIdx.x] = s;
}

#define ACC_CLOBBERS "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79", \
 "v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95", \
 "v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111", \
 "v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127"

__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_vgpr(64)))
void bench_regacc(const int *__restrict__ rows, int per_wave, float *__restrict__ out)
{
    extern __shared__ float acc[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < kRows * 64; i += 1024) acc[i] = 1.f;
    __syncthreads();
    for (int i = 0; i < 64; i++) {
        const int ii = __builtin_amdgcn_readfirstlane(i);
        asm volatile("s_set_gpr_idx_on %0, gpr_idx(DST)\n\tv_mov_b32 v64, 0\n\ts_set_gpr_idx_off" :: "s"(ii) : ACC_CLOBBERS);
    }
    const int *mine = rows + ((size_t)blockIdx.x * 16 + wave) * per_wave;
    for (int i = 0; i < per_wave; i += 8) {
        float t[8];
        int d[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int r = __builtin_amdgcn_readfirstlane(mine[i + k]);
            d[k] = r & 63;                                     // destination register of this edge
            t[k] = acc[r * 64 + lane];                         // its source row, staged in LDS
        }
        asm volatile("s_set_gpr_idx_on %8, gpr_idx(SRC0,DST)\n\tv_add_f32 v64, v64, %0\n\t"
                     "s_set_gpr_idx_idx %9\n\tv_add_f32 v64, v64, %1\n\t"
                     "s_set_gpr_idx_idx %10\n\tv_add_f32 v64, v64, %2\n\t"
                     "s_set_gpr_idx_idx %11\n\tv_add_f32 v64, v64, %3\n\t"
                     "s_set_gpr_idx_idx %12\n\tv_add_f32 v64, v64, %4\n\t"
                     "s_set_gpr_idx_idx %13\n\tv_add_f32 v64, v64, %5\n\t"
                     "s_set_gpr_idx_idx %14\n\tv_add_f32 v64, v64, %6\n\t"
                     "s_set_gpr_idx_idx %15\n\tv_add_f32 v64, v64, %7\n\ts_set_gpr_idx_off"
                     :: "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]), "v"(t[4]), "v"(t[5]), "v"(t[6]), "v"(t[7]),
                        "s"(d[0]), "s"(d[1]), "s"(d[2]), "s"(d[3]), "s"(d[4]), "s"(d[5]), "s"(d[6]), "s"(d[7]) : ACC_CLOBBERS);
    }
    float s = 0.f;
    for (int i = 0; i < 64; i++) {
        const int ii = __builtin_amdgcn_readfirstlane(i);
        float v;
        asm volatile("s_set_gpr_idx_on %1, gpr_idx(SRC0)\n\tv_mov_b32 %0, v64\n\ts_set_gpr_idx_off" : "=v"(v) : "s"(ii) : ACC_CLOBBERS);
        s += v;
    }
    // every row op added 1.0 per lane: the total over the 64 registers must equal per_wave exactly (fp32-exact below 2^24)
    if (s == (float)per_wave && lane == 0) atomicAdd(&out[blockIdx.x], 1.f);       // 16 exact wavefronts per workgroup
}

int main(int argc, char **argv)
{
    const int per_wave = argc > 1 ? atoi(argv[1]) : 16384;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    std::vector<int> h((size_t)cus * 16 * per_wave);
    unsigned s = 12345;
    for (auto &x : h) { s = s * 1664525u + 1013904223u; x = (int)((s >> 10) % kRows); }
    int *d; float *o;
    CHECK(hipMalloc(&d, h.size() * sizeof(int)));
    CHECK(hipMalloc(&o, cus * sizeof(float)));
    CHECK(hipMemcpy(d, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice));
    const size_t lds = (size_t)kRows * 256;
    auto run = [&](const char *name, auto kern) {
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipEvent_t a, b;
        CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        float best = 1e9f;
        for (int rep = 0; rep < 4; rep++) {
            CHECK(hipEventRecord(a));
            hipLaunchKernelGGL(kern, dim3(cus), dim3(1024), lds, 0, d, per_wave, o);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
        }
        const double ops_per_cu = 16.0 * per_wave;
        printf("%-44s %8.3f ms  %6.2f ns per row-op per CU  = %5.2f cycles at 2.4 GHz (%.1f at 2.0)\n", name, best,
               best * 1e6 / ops_per_cu, best * 1e6 / ops_per_cu * 2.4, best * 1e6 / ops_per_cu * 2.0);
    };
    run("mode 0  ds_add_f32 (no return)", bench<0>);
    run("mode 1  ds_read_b32 + add + ds_write_b32", bench<1>);
    run("mode 2  ds_read_addtid + add + ds_write_addtid", bench<2>);
    run("mode 3  ds_read_b32 + wait", bench<3>);
    run("mode 4  ds_add_rtn_f32 + wait", bench<4>);
    run("mode 5  8 x ds_read_b32 in flight + adds", bench_read8);
    CHECK(hipMemset(o, 0, cus * sizeof(float)));
    run("mode 6  8 x ds_read_b32 + 8 indexed VGPR adds", bench_regacc);
    std::vector<float> ho(cus);
    CHECK(hipMemcpy(ho.data(), o, cus * sizeof(float), hipMemcpyDeviceToHost));
    int good = 0;
    for (float x : ho) good += x == 16.f * 4;     // (4 timed launches)
    printf("mode 6 register-indexed sums exact on %d of %d workgroups\n", good, cus);
    return 0;
}
