// gather_ceiling.hip -- measurement tool, not part of the library: what does the bare ACCESS STREAM of the sliced
// aggregation cost on this chip?  The kernel walks a prepared list of source-row ids (the ids in the order the
// streaming kernel consumes them: phase-major, then edge order) in segments of `seg` ids per wavefront, keeps U
// wave-wide row loads (64 / LPR rows each) in flight, sums them and writes one vector per wavefront.  No partition,
// no pieces, no flushes: the time of this kernel is the floor any schedule with the same slices and the same
// dispatch order has to live with.  D = 4 * LPR floats per row.
#include <hip/hip_runtime.h>
#include <cstdint>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int LPR, int U>
__global__ void __launch_bounds__(256) gather_ceiling(const float *__restrict__ X, const int32_t *__restrict__ ids,
                                                      int64_t n, int seg, float *__restrict__ out)
{
    constexpr int RPI = 64 / LPR;                 // rows per wave-wide load
    constexpr int LOADS = 64 / RPI;               // loads per tile of 64 ids
    static_assert(LOADS % U == 0, "");
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t base = wave * (int64_t)seg;
    if (base + seg > n) return;
    const int lslot = lane / LPR, c = lane % LPR;
    const char *xb = reinterpret_cast<const char *>(X) + c * 16;
    const uint32_t row_bytes = LPR * 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int idv = ids[base + lane];
    for (int t = 0; t < seg; t += 64) {
        const int idn = (t + 64 < seg) ? ids[base + t + 64 + lane] : 0;
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t id = (uint32_t)__shfl(idv, u * RPI + lslot);
            v[u] = *reinterpret_cast<const f32x4 *>(xb + id * row_bytes);
        }
#pragma unroll
        for (int b = 1; b < LOADS / U; b++) {
            uint32_t nn[U];
#pragma unroll
            for (int u = 0; u < U; u++) nn[u] = (uint32_t)__shfl(idv, (b * U + u) * RPI + lslot) * row_bytes;
#pragma unroll
            for (int u = 0; u < U; u++) {
                acc += v[u];
                v[u] = *reinterpret_cast<const f32x4 *>(xb + nn[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u];
        idv = idn;
    }
    reinterpret_cast<f32x4 *>(out)[wave * 64 + lane] = acc;
}

// Variant with a cache of hot rows in LDS: 1024-thread workgroups (one per CU: 16 wavefronts), `hub_rows` rows of
// the table `hub` copied to LDS up front; an id with the top bit set is "0x80000000 | slot" and is read from LDS.
// (hub_rows = 0: the same kernel shape without the cache -- the floor at 16 wavefronts per CU.)
template <int LPR, int U>
__global__ void __launch_bounds__(1024) gather_ceiling_hub(const float *__restrict__ X, const int32_t *__restrict__ ids,
                                                           int64_t n, int seg, float *__restrict__ out,
                                                           const float *__restrict__ hub, int hub_rows, int lds_bytes_used)
{
    constexpr int RPI = 64 / LPR;
    constexpr int LOADS = 64 / RPI;
    extern __shared__ float s_hub[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < hub_rows * LPR * 4; i += 1024) s_hub[i] = hub[i];
    __syncthreads();
    const int wib = threadIdx.x >> 6;
    const int lslot = lane / LPR, c = lane % LPR;
    const char *xb = reinterpret_cast<const char *>(X) + c * 16;
    const char *hb = reinterpret_cast<const char *>(s_hub) + c * 16;
    const uint32_t row_bytes = LPR * 16;
    const int64_t waves_total = (int64_t)gridDim.x * 16;
    const int64_t nseg = n / seg;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // persistent: wavefront w takes segments w, w + W, ... (phase-major order is kept chip-wide)
    for (int64_t sgi = (int64_t)blockIdx.x * 16 + wib; sgi < nseg; sgi += waves_total) {
        const int64_t base = sgi * (int64_t)seg;
        int idv = ids[base + lane];
        for (int t = 0; t < seg; t += 64) {
            const int idn = (t + 64 < seg) ? ids[base + t + 64 + lane] : 0;
            auto load = [&](uint32_t id) -> f32x4 {
                if (id >> 31) return *reinterpret_cast<const f32x4 *>(hb + (id & 0x7fffffffu) * row_bytes);
                return *reinterpret_cast<const f32x4 *>(xb + id * row_bytes);
            };
            f32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; u++) v[u] = load((uint32_t)__shfl(idv, u * RPI + lslot));
#pragma unroll
            for (int b = 1; b < LOADS / U; b++) {
                uint32_t nn[U];
#pragma unroll
                for (int u = 0; u < U; u++) nn[u] = (uint32_t)__shfl(idv, (b * U + u) * RPI + lslot);
#pragma unroll
                for (int u = 0; u < U; u++) {
                    acc += v[u];
                    v[u] = load(nn[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++) acc += v[u];
            idv = idn;
        }
    }
    reinterpret_cast<f32x4 *>(out)[((int64_t)blockIdx.x * 16 + wib) * 64 + lane] = acc;
}

// Which XCD does block b run on?  out[b] = XCC_ID hardware register of the block's first wavefront (the library's
// "block b lands on XCD b % 8" is a locality assumption; this checks it on the device at hand).
__global__ void xcc_of_block(int32_t *__restrict__ out)
{
    if (threadIdx.x == 0) {
        uint32_t v;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
        out[blockIdx.x] = (int32_t)(v & 0xf);
    }
}

// Variant: an id with the top bit set marks a COLD row -- it is loaded with a non-temporal load (first to leave the L2),
// so that the hot rows of the slice keep their lines.  Same walk as gather_ceiling.
template <int LPR, int U>
__global__ void __launch_bounds__(256) gather_ceiling_nt(const float *__restrict__ X, const int32_t *__restrict__ ids,
                                                         int64_t n, int seg, float *__restrict__ out)
{
    constexpr int RPI = 64 / LPR;
    constexpr int LOADS = 64 / RPI;
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t base = wave * (int64_t)seg;
    if (base + seg > n) return;
    const int lslot = lane / LPR, c = lane % LPR;
    const char *xb = reinterpret_cast<const char *>(X) + c * 16;
    const uint32_t row_bytes = LPR * 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // buffer loads, so that the cache policy is an operand of the instruction the compiler cannot fold away (a plain and a
    // __builtin_nontemporal_load in the two arms of a branch are merged into ONE plain load): aux 0 = default, 2 = nt
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(X), (short)0, 0x7fffffff, 0x00020000);
    auto load = [&](uint32_t id) -> f32x4 {
        const int off = (int)((id & 0x7fffffffu) * row_bytes + (uint32_t)c * 16u);
        i32x4_t r;
        if (id >> 31) r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 2);
        else r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        return __builtin_bit_cast(f32x4, r);
    };
    int idv = ids[base + lane];
    for (int t = 0; t < seg; t += 64) {
        const int idn = (t + 64 < seg) ? ids[base + t + 64 + lane] : 0;
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = load((uint32_t)__shfl(idv, u * RPI + lslot));
#pragma unroll
        for (int b = 1; b < LOADS / U; b++) {
            uint32_t nn[U];
#pragma unroll
            for (int u = 0; u < U; u++) nn[u] = (uint32_t)__shfl(idv, (b * U + u) * RPI + lslot);
#pragma unroll
            for (int u = 0; u < U; u++) {
                acc += v[u];
                v[u] = load(nn[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u];
        idv = idn;
    }
    reinterpret_cast<f32x4 *>(out)[wave * 64 + lane] = acc;
}

extern "C" __attribute__((visibility("default")))
int gather_ceiling_nt_launch(const float *X, const int32_t *ids, int64_t n, int dim, int seg, float *out)
{
    if (seg % 64 != 0 || seg <= 0 || dim != 64) return -1;
    const int64_t waves = n / seg;
    const unsigned grid = (unsigned)((waves + 3) / 4);
    if (grid == 0) return 0;
    hipLaunchKernelGGL((gather_ceiling_nt<16, 4>), dim3(grid), dim3(256), 0, 0, X, ids, n, seg, out);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

extern "C" __attribute__((visibility("default")))
int xcc_of_block_launch(int32_t *out, int blocks, int threads)
{
    hipLaunchKernelGGL(xcc_of_block, dim3(blocks), dim3(threads), 0, 0, out);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// How many 256-byte row flushes per second does the chip take?  Every wavefront makes `per_wave` flushes of one
// 64-float row each to pseudo-random rows of Y[rows][64]; how: 0 float atomics (what the sliced schedule's flush does),
// 1 plain stores, 2 plain read-add-write.
__global__ void __launch_bounds__(256) flush_rate(float *__restrict__ Y, int64_t rows, int per_wave, int how)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint64_t h = wave * 0x9E3779B97F4A7C15ull + 12345;
    for (int k = 0; k < per_wave; k++) {
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        float *dst = Y + (h % (uint64_t)rows) * 64 + lane;
        if (how == 0) unsafeAtomicAdd(dst, 1.0f);
        else if (how == 1) *dst = 1.0f;
        else *dst = *dst + 1.0f;
    }
}

extern "C" __attribute__((visibility("default")))
int flush_rate_launch(float *Y, int64_t rows, int64_t waves, int per_wave, int how)
{
    hipLaunchKernelGGL(flush_rate, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, 0, Y, rows, per_wave, how);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

extern "C" __attribute__((visibility("default")))
int gather_ceiling_hub_launch(const float *X, const int32_t *ids, int64_t n, int dim, int seg, int U, float *out,
                              const float *hub, int hub_rows, int lds_bytes, int blocks)
{
    if (seg % 64 != 0 || seg <= 0 || (dim != 64 && dim != 32 && dim != 16)) return -1;
    if ((size_t)hub_rows * dim * 4 > (size_t)lds_bytes) return -4;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void *)gather_ceiling_hub<16, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)gather_ceiling_hub<16, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)gather_ceiling_hub<8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)gather_ceiling_hub<8, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)gather_ceiling_hub<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)gather_ceiling_hub<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
#define HUB(L, UU) hipLaunchKernelGGL((gather_ceiling_hub<L, UU>), dim3(blocks), dim3(1024), lds_bytes, 0, X, ids, n, seg, out, hub, hub_rows, lds_bytes)
    if (dim == 64 && U == 4) HUB(16, 4);
    else if (dim == 64 && U == 8) HUB(16, 8);
    else if (dim == 32 && U == 4) HUB(8, 4);
    else if (dim == 32 && U == 8) HUB(8, 8);
    else if (dim == 16 && U == 4) HUB(4, 4);
    else if (dim == 16 && U == 2) HUB(4, 2);
    else return -2;
#undef HUB
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

extern "C" __attribute__((visibility("default")))
int gather_ceiling_launch(const float *X, const int32_t *ids, int64_t n, int dim, int seg, int U, float *out)
{
    if (seg % 64 != 0 || seg <= 0) return -1;
    const int64_t waves = n / seg;
    const unsigned grid = (unsigned)((waves + 3) / 4);
    if (grid == 0) return 0;
#define GO(L, UU) hipLaunchKernelGGL((gather_ceiling<L, UU>), dim3(grid), dim3(256), 0, 0, X, ids, n, seg, out)
    if (dim == 64 && U == 4) GO(16, 4);
    else if (dim == 64 && U == 8) GO(16, 8);
    else if (dim == 64 && U == 16) GO(16, 16);
    else if (dim == 128 && U == 4) GO(32, 4);
    else if (dim == 128 && U == 8) GO(32, 8);
    else if (dim == 32 && U == 4) GO(8, 4);
    else if (dim == 32 && U == 8) GO(8, 8);
    else if (dim == 16 && U == 4) GO(4, 4);
    else if (dim == 16 && U == 2) GO(4, 2);
    else return -2;
#undef GO
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// SDDMM floor (round 6): the same walk as gather_ceiling -- the ids phase-major, U row loads in flight -- but every gathered row
// is multiplied with a destination row's piece held in REGISTERS for the whole segment (the best any schedule can do for the
// destination side), reduced over the LPR lanes of its slot, and the 64 dot products of a tile leave as ONE coalesced 256-byte
// store (edge_out in the order the ids are consumed: the best any schedule can do for the output).  What is left is the
// gather stream + 4 bytes per edge: the floor of edge_out[e] = <A[row(e)], X[col(e)]> on these slices.
template <int LPR, int U>
__global__ void __launch_bounds__(256) sddmm_ceiling(const float *__restrict__ X, const int32_t *__restrict__ ids,
                                                     int64_t n, int seg, const float *__restrict__ A, int64_t a_rows,
                                                     float *__restrict__ edge_out)
{
    constexpr int RPI = 64 / LPR;
    constexpr int LOADS = 64 / RPI;
    static_assert(LOADS % U == 0, "");
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t base = wave * (int64_t)seg;
    if (base + seg > n) return;
    const int lslot = lane / LPR, c = lane % LPR;
    const char *xb = reinterpret_cast<const char *>(X) + c * 16;
    const uint32_t row_bytes = LPR * 16;
    const f32x4 a = *reinterpret_cast<const f32x4 *>(A + (wave % a_rows) * (int64_t)(LPR * 4) + c * 4);
    int idv = ids[base + lane];
    for (int t = 0; t < seg; t += 64) {
        const int idn = (t + 64 < seg) ? ids[base + t + 64 + lane] : 0;
        float mine = 0.f;                      // lane l ends up with the dot product of the tile's id l
        auto finish = [&](f32x4 v, int load) {
            const f32x4 pr = v * a;
            float d = (pr.x + pr.y) + (pr.z + pr.w);
#pragma unroll
            for (int w = LPR / 2; w > 0; w >>= 1) d += __shfl_xor(d, w);      // every lane of the slot holds the slot's dot
            // id index of (load, slot) = load * RPI + slot: lane l wants load l / RPI, slot l % RPI
            const float got = __shfl(d, (lane % RPI) * LPR);
            if (lane / RPI == load) mine = got;
        };
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t id = (uint32_t)__shfl(idv, u * RPI + lslot);
            v[u] = *reinterpret_cast<const f32x4 *>(xb + id * row_bytes);
        }
#pragma unroll
        for (int b = 1; b < LOADS / U; b++) {
            uint32_t nn[U];
#pragma unroll
            for (int u = 0; u < U; u++) nn[u] = (uint32_t)__shfl(idv, (b * U + u) * RPI + lslot) * row_bytes;
#pragma unroll
            for (int u = 0; u < U; u++) {
                finish(v[u], (b - 1) * U + u);
                v[u] = *reinterpret_cast<const f32x4 *>(xb + nn[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) finish(v[u], (LOADS / U - 1) * U + u);
        __builtin_nontemporal_store(mine, edge_out + base + t + lane);
        idv = idn;
    }
}

extern "C" __attribute__((visibility("default")))
int sddmm_ceiling_launch(const float *X, const int32_t *ids, int64_t n, int dim, int seg, int U, const float *A, int64_t a_rows,
                         float *edge_out)
{
    if (seg % 64 != 0 || seg <= 0 || dim != 64) return -1;
    const int64_t waves = n / seg;
    const unsigned grid = (unsigned)((waves + 3) / 4);
    if (grid == 0) return 0;
    if (U == 4) hipLaunchKernelGGL((sddmm_ceiling<16, 4>), dim3(grid), dim3(256), 0, 0, X, ids, n, seg, A, a_rows, edge_out);
    else if (U == 8) hipLaunchKernelGGL((sddmm_ceiling<16, 8>), dim3(grid), dim3(256), 0, 0, X, ids, n, seg, A, a_rows, edge_out);
    else return -2;
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
