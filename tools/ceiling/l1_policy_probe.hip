// l1_policy_probe.hip -- measurement tool, not part of the library.  Question: the sliced gather makes ~2 L2 requests per
// edge at D = 64 with ~0 % reuse in the CU's 32 KB vector L1 (TCP) although the most gathered rows of a slice take a fifth
// of its edges -- the stream of cold rows evicts them.  Can the L1 be kept for the hot rows by loading the COLD rows with a
// cache policy that bypasses it?  (MI355X_MICROARCH.md: `sc1` / `sc0 sc1` loads are L2-served and bypass the L1 only; `nt`
// also demotes the line in the L2 -- measured in round 3: far slower.)
//
// Same walk as gather_ceiling (tools/ceiling/gather_ceiling.hip): a wavefront streams `seg` ids, U wave-wide row loads in
// flight, sums, writes one vector.  An id with the top bit set is COLD; a load instruction's policy is decided by the first
// row it fetches (policy is per instruction; the harness arranges hot ids first inside every segment, so all but one load
// of a segment are uniform).  AUX is the buffer-load cache-policy operand for cold rows: 0 plain, 1 sc0, 16 sc1, 17 sc0 sc1.
// Rows are `row_bytes` apart (256: contiguous 64-float rows; 512: the library's gapped layout).
#include <hip/hip_runtime.h>
#include <cstdint>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));

template <int AUX, int U>
__global__ void __launch_bounds__(256) l1_policy_walk(const float *__restrict__ X, const int32_t *__restrict__ ids, int64_t n,
                                                      int seg, uint32_t row_bytes, float *__restrict__ out)
{
    constexpr int LPR = 16, RPI = 4, LOADS = 16;      // D = 64: 16 lanes per row, 4 rows per wave-wide load
    static_assert(LOADS % U == 0, "");
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t base = wave * (int64_t)seg;
    if (base + seg > n) return;
    const int lslot = lane / LPR, c = lane % LPR;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(X), (short)0, 0x7fffffff, 0x00020000);
    auto load = [&](uint32_t id, uint32_t first_id) -> f32x4 {
        const int off = (int)((id & 0x7fffffffu) * row_bytes + (uint32_t)c * 16u);
        i32x4_t r;
        if (__builtin_amdgcn_readfirstlane(first_id) >> 31) r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, AUX);
        else r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        return __builtin_bit_cast(f32x4, r);
    };
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int idv = ids[base + lane];
    for (int t = 0; t < seg; t += 64) {
        const int idn = (t + 64 < seg) ? ids[base + t + 64 + lane] : 0;
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = load((uint32_t)__shfl(idv, u * RPI + lslot), (uint32_t)__shfl(idv, u * RPI));
#pragma unroll
        for (int b = 1; b < LOADS / U; b++) {
            uint32_t nn[U], nf[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                nn[u] = (uint32_t)__shfl(idv, (b * U + u) * RPI + lslot);
                nf[u] = (uint32_t)__shfl(idv, (b * U + u) * RPI);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                acc += v[u];
                v[u] = load(nn[u], nf[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u];
        idv = idn;
    }
    reinterpret_cast<f32x4 *>(out)[wave * 64 + lane] = acc;
}

extern "C" __attribute__((visibility("default")))
int l1_policy_launch(const float *X, const int32_t *ids, int64_t n, int seg, int row_bytes, int aux, float *out)
{
    if (seg % 64 != 0 || seg <= 0 || (row_bytes != 256 && row_bytes != 512)) return -1;
    const int64_t waves = n / seg;
    const unsigned grid = (unsigned)((waves + 3) / 4);
    if (grid == 0) return 0;
#define GO(A) hipLaunchKernelGGL((l1_policy_walk<A, 8>), dim3(grid), dim3(256), 0, 0, X, ids, n, seg, (uint32_t)row_bytes, out)
    if (aux == 0) GO(0);
    else if (aux == 1) GO(1);
    else if (aux == 16) GO(16);
    else if (aux == 17) GO(17);
    else if (aux == 2) GO(2);
    else return -2;
#undef GO
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
