// gnna_gemm.hip -- the one dense product of the GCN / GIN layers that the BLAS library handles badly:
//
//     dW[K, N] = X^T[K, M] * G[M, N]        (M = number of nodes, 1e5..1e8;  K, N = feature widths)
//
// i.e. the weight gradient of spmm_backward_cuda / spmm_backward_cuda_gin (reference
// GNNAdvisor_kernel.cu:473, 710: torch::mm(X.transpose(0,1), d_input_prime)).  It is a reduction
// over the node dimension of M rank-1 updates; rocBLAS/hipBLASLt run it at 3.0 ms for
// M = 2.45 M, K = N = 64 (products-like) where reading X and G once takes 0.21 ms.
//
// Kernel: a 256-thread block owns a 64 x 64 tile of dW and a slab of rows.  Each wavefront walks
// its share of the slab four rows at a time: lane (s, q) = (lane / 16, lane % 16) loads
// X[row + s, k0 + 4q .. +3] and G[row + s, n0 + 4q .. +3] with one dwordx4 each (4 rows x 256 B
// per wave-wide load), and the 4 x 4 component pairs feed 16 v_mfma_f32_16x16x4_f32
// (A[i = q][kk = s] = X component c, B[kk = s][j = q] = G component c'): the MFMA's reduction
// dimension is the node dimension.  The 16 accumulator tiles (64 VGPRs) hold
// dW[k0 + 4 i + c][n0 + 4 j + c'].  Wavefronts of a block are summed in LDS in a fixed order, every
// block writes its 64 x 64 partial to scratch, a second kernel sums the slabs (deterministic; no
// global atomics).  fp32 in, fp32 MFMA, fp32 accumulate.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "gnna.h"
#include "gnna_device.h"
#include "gnna_internal.h"

namespace gnna {
namespace {

typedef VecOf<4>::T f32x4;
typedef VecOf<4>::M f32x4_mem;
constexpr int kTile = 64;

// 4 consecutive floats at p (= row start + col); zeros for columns >= ncols and when !row_ok
__device__ __forceinline__ f32x4 load4(const float *__restrict__ p, int col, int ncols, bool row_ok)
{
    f32x4 v = (f32x4)(0.f);
    if (row_ok && col < ncols) {
        if (col + 4 <= ncols) {
            v = *reinterpret_cast<const f32x4_mem *>(p);
        } else {
            v[0] = p[0];
            if (col + 1 < ncols) v[1] = p[1];
            if (col + 2 < ncols) v[2] = p[2];
        }
    }
    return v;
}

__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 4)))
xtg_kernel(const float *__restrict__ X, const float *__restrict__ G, float *__restrict__ part,
           int64_t M, int K, int N, int64_t rows_per_block, int kblocks, int nblocks)
{
    __shared__ float tile[kTile * kTile];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Hardware deals consecutive blocks to the 8 XCDs in turn; the tiles of one slab read the same rows
    // of X and G, so they are placed on one XCD (one L2): block x + 8 y is the y-th (slab, tile) pair of
    // XCD x.  Slabs beyond the last full round of 8 keep the plain order.
    const int64_t tiles = (int64_t)nblocks * kblocks;
    int64_t bid = blockIdx.x;
    {
        const int64_t full = (int64_t)gridDim.x / (kXcds * tiles) * (kXcds * tiles);
        if (bid < full) {
            const int64_t x = bid % kXcds, y = bid / kXcds;
            bid = ((y / tiles) * kXcds + x) * tiles + y % tiles;
        }
    }
    const int nb = (int)(bid % nblocks);
    const int kb = (int)((bid / nblocks) % kblocks);
    const int64_t slab = bid / tiles;
    const int64_t m_begin = slab * rows_per_block;
    const int64_t m_end = m_begin + rows_per_block < M ? m_begin + rows_per_block : M;
    const int s = lane >> 4, q = lane & 15;
    const int kc = kb * kTile + 4 * q, nc = nb * kTile + 4 * q;

    f32x4 acc[4][4];
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int d = 0; d < 4; d++) acc[c][d] = (f32x4)(0.f);

    // 4 waves x 4 rows per step: this lane's rows are lr, lr + 16, ... of the slab.  Software pipeline:
    // the two loads of the next step are issued before the 16 MFMAs of the current one.  Pointers walk
    // by 16 rows; everything is relative to the slab so that the row index stays a small int.
    const int nrows = (int)(m_end - m_begin);
    int lr = wave * 4 + s;
    const float *xp = X + (m_begin + lr) * (int64_t)K + kc;
    const float *gp = G + (m_begin + lr) * (int64_t)N + nc;
    const int64_t xstep = (int64_t)16 * K, gstep = (int64_t)16 * N;
    const int steps = (nrows - wave * 4 + 15) / 16;   // wave-uniform: steps in which this wave has at least one row
    // kPrefetch steps are in flight: the loads of step it + kPrefetch are issued right after the MFMAs
    // of step it have consumed their registers
    constexpr int kPrefetch = 3;
    f32x4 a[kPrefetch], b[kPrefetch];
#pragma unroll
    for (int j = 0; j < kPrefetch; j++) {
        a[j] = load4(xp, kc, K, lr < nrows);
        b[j] = load4(gp, nc, N, lr < nrows);
        xp += xstep; gp += gstep; lr += 16;
    }
    for (int it = 0; it < steps; it += kPrefetch) {
#pragma unroll
        for (int j = 0; j < kPrefetch; j++) {
            if (it + j < steps) {
#pragma unroll
                for (int c = 0; c < 4; c++)
#pragma unroll
                    for (int d = 0; d < 4; d++)
                        acc[c][d] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][c], b[j][d], acc[c][d], 0, 0, 0);
                a[j] = load4(xp, kc, K, lr < nrows);
                b[j] = load4(gp, nc, N, lr < nrows);
                xp += xstep; gp += gstep; lr += 16;
            }
        }
    }

    // D layout of v_mfma_f32_16x16x4_f32: lane (s, q), register r holds D[i = 4 s + r][j = q].
    // The four wavefronts are summed into the LDS tile one after the other (fixed order: the
    // result is bit-reproducible); within a wavefront every lane owns distinct tile elements.
    for (int w = 0; w < kWavesPerBlock; w++) {
        if (wave == w) {
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int d = 0; d < 4; d++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int kk = 4 * (4 * s + r) + c, nn = 4 * q + d;
                        if (w == 0) tile[kk * kTile + nn] = acc[c][d][r];
                        else tile[kk * kTile + nn] += acc[c][d][r];
                    }
        }
        __syncthreads();
    }
    float *dst = part + bid * (int64_t)(kTile * kTile);
    for (int i = threadIdx.x; i < kTile * kTile; i += kBlock) dst[i] = tile[i];
}

// dW[k][n] = sum over slabs of the partial tiles.  A block owns 16 consecutive elements of one tile
// row; thread (e, g) = (tid % 16, tid / 16) sums slabs g, g + 16, ... (independent loads, 64-byte
// coalesced), the 16 partial sums of an element are folded in LDS.
__global__ void __launch_bounds__(kBlock)
xtg_reduce_kernel(const float *__restrict__ part, float *__restrict__ dW, int N, int64_t slabs,
                  int kblocks, int nblocks)
{
    __shared__ float fold[16][17];
    const int e = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int segs_per_row = (N + 15) / 16;
    const int k = (int)(blockIdx.x / segs_per_row);
    const int n = (int)(blockIdx.x % segs_per_row) * 16 + e;
    float acc = 0.f;
    if (n < N) {
        const int kb = k / kTile, nb = n / kTile;
        const int64_t slab_stride = (int64_t)kblocks * nblocks * (kTile * kTile);
        const float *p = part + ((int64_t)kb * nblocks + nb) * (kTile * kTile) + (k % kTile) * kTile + (n % kTile);
#pragma unroll 4
        for (int64_t sl = g; sl < slabs; sl += 16) acc += p[sl * slab_stride];
    }
    fold[g][e] = acc;
    __syncthreads();
    if (g == 0 && n < N) {
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 16; j++) sum += fold[j][e];
        dW[(int64_t)k * N + n] = sum;
    }
}

}  // namespace
}  // namespace gnna

using namespace gnna;

extern "C" {
#pragma GCC visibility push(default)

int gnna_xtg_f32(const float *X, const float *G, float *dW, int64_t num_rows, int K, int N, void *stream_v)
{
    if (num_rows < 0 || K < 0 || N < 0) return fail(GNNA_ERR_INVALID_ARGUMENT, "negative size");
    if (K == 0 || N == 0) return GNNA_OK;
    if (!dW || (num_rows > 0 && (!X || !G))) return fail(GNNA_ERR_INVALID_ARGUMENT, "null pointer");
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    DeviceState *ds = nullptr;
    int rc = get_device_state(&ds);
    if (rc != GNNA_OK) return rc;
    if (num_rows == 0) {
        hipError_t e = hipMemsetAsync(dW, 0, (size_t)K * N * sizeof(float), stream);
        return e == hipSuccess ? GNNA_OK : fail(GNNA_ERR_HIP, "memset: %s", hipGetErrorString(e));
    }
    const int kblocks = (K + kTile - 1) / kTile, nblocks = (N + kTile - 1) / kTile;
    const int64_t tiles = (int64_t)kblocks * nblocks;
    // four blocks (16 wavefronts) per CU, at least 256 rows each (64 on small graphs)
    int64_t slabs = std::max<int64_t>(1, ((int64_t)ds->num_cus * 4 + tiles - 1) / tiles);
    const int64_t min_rows = num_rows >= 65536 ? 256 : 64;
    slabs = std::min<int64_t>(slabs, (num_rows + min_rows - 1) / min_rows);
    int64_t rows_per_block = ((num_rows + slabs - 1) / slabs + 15) / 16 * 16;
    slabs = (num_rows + rows_per_block - 1) / rows_per_block;
    const int64_t blocks = slabs * tiles;
    if (blocks > 0x7fffffff) return fail(GNNA_ERR_INVALID_ARGUMENT, "problem too large");
    void *ws = nullptr;
    rc = get_workspace(ds, stream, 2, (size_t)blocks * kTile * kTile * sizeof(float), &ws);
    if (rc != GNNA_OK) return rc;
    hipLaunchKernelGGL(xtg_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, stream, X, G, static_cast<float *>(ws),
                       num_rows, K, N, rows_per_block, kblocks, nblocks);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GNNA_ERR_HIP, "xtg launch: %s", hipGetErrorString(e));
    const int64_t rblocks = (int64_t)K * ((N + 15) / 16);
    hipLaunchKernelGGL(xtg_reduce_kernel, dim3((unsigned)rblocks), dim3(kBlock), 0, stream,
                       static_cast<const float *>(ws), dW, N, slabs, kblocks, nblocks);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(GNNA_ERR_HIP, "xtg reduce launch: %s", hipGetErrorString(e));
    return GNNA_OK;
}

#pragma GCC visibility pop
}  // extern "C"
