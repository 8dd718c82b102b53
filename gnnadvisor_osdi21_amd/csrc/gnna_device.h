// gnna_device.h -- device-side building blocks shared by the kernels of libgnna.so
// (lane layout constants, vector register/memory types, wave-level folds).
#ifndef GNNA_DEVICE_H_
#define GNNA_DEVICE_H_

#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

namespace gnna {


constexpr int kWave = 64;
constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kXcds = 8;

enum { MODE_SAG = 0, MODE_GCN = 1, MODE_GIN = 2, MODE_SDDMM = 3 };

// T: register type; M: the same vector as it sits in memory.  Feature rows are only 4-byte
// aligned in general (row stride = D floats, D arbitrary), and gfx950 global_load/store_dwordx4
// need no more than dword alignment, so M is declared with alignment 4.
template <int VEC> struct VecOf;
template <> struct VecOf<1> { typedef float T; typedef float M; };
template <> struct VecOf<4> {
    typedef float T __attribute__((ext_vector_type(4)));
    typedef T M __attribute__((aligned(4)));
};

// ---- wave-level helpers ----------------------------------------------------------------

__device__ __forceinline__ float fold_xor16(float v)
{
    unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ __forceinline__ float fold_xor32(float v)
{
    unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// DPP row rotation by N lanes inside each 16-lane row (pure VALU, folds into the add).
template <int N>
__device__ __forceinline__ float row_ror(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xF, 0xF, false));
}

// Sum over the 64/LPR lanes that share lane % LPR; result in every lane.
template <int LPR>
__device__ __forceinline__ float slot_reduce(float v)
{
    if constexpr (LPR <= 8) v += row_ror<8>(v);   // strides 8 and 4 stay inside a 16-lane DPP row
    if constexpr (LPR <= 4) v += row_ror<4>(v);
    if constexpr (LPR <= 16) v = fold_xor16(v);
    if constexpr (LPR <= 32) v = fold_xor32(v);
    return v;
}

// DPP data movement inside a 16-lane row.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}

// Sum over the LPR consecutive lanes of a slot (lanes sharing lane / LPR); result in every lane of the slot.
template <int LPR>
__device__ __forceinline__ float lane_group_sum(float v)
{
    v += dpp_move<0xB1>(v);                          // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);                          // quad_perm [2,3,0,1]
    if constexpr (LPR >= 8) v += dpp_move<0x141>(v);  // row_half_mirror: the other quad of the 8
    if constexpr (LPR >= 16) v += dpp_move<0x140>(v); // row_mirror: the other half of the 16
    if constexpr (LPR >= 32) v = fold_xor16(v);
    if constexpr (LPR >= 64) v = fold_xor32(v);
    return v;
}

template <int VEC>
__device__ __forceinline__ typename VecOf<VEC>::T vzero()
{
    typename VecOf<VEC>::T z;
    if constexpr (VEC == 1) z = 0.f; else z = (typename VecOf<VEC>::T)(0.f);
    return z;
}

template <int VEC>
__device__ __forceinline__ float vget(const typename VecOf<VEC>::T &v, int k)
{
    if constexpr (VEC == 1) return v; else return v[k];
}

template <int VEC>
__device__ __forceinline__ void vset(typename VecOf<VEC>::T &v, int k, float x)
{
    if constexpr (VEC == 1) v = x; else v[k] = x;
}

// Inclusive prefix sum over the 64 lanes with DPP row shifts and row broadcasts (no LDS round trips).
__device__ __forceinline__ int wave_inclusive_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
    return v;
}


// fold_row: folds the RPI slots of `acc`.  LPR <= 16: reduce-scatter with v_permlane32_swap /
// v_permlane16_swap (+ DPP rotations); every lane ends with ONE float, component lane>>4 of piece lane%LPR
// (returned in [0]).  Wider rows: butterfly per component, every slot ends with the row's 4-float piece.
template <int LPR, int MODE>
__device__ __forceinline__ typename VecOf<4>::T fold_row(const typename VecOf<4>::T acc, float scale)
{
    typedef typename VecOf<4>::T VT;
    VT r = acc;
    if constexpr (LPR <= 16) {
        float px, qy;
        {
            auto t = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[0]), __float_as_uint(acc[2]), false, false);
            px = __uint_as_float(t[0]) + __uint_as_float(t[1]);
            t = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[1]), __float_as_uint(acc[3]), false, false);
            qy = __uint_as_float(t[0]) + __uint_as_float(t[1]);
        }
        float val;
        {
            auto t = __builtin_amdgcn_permlane16_swap(__float_as_uint(px), __float_as_uint(qy), false, false);
            val = __uint_as_float(t[0]) + __uint_as_float(t[1]);
        }
        if constexpr (LPR <= 8) val += row_ror<8>(val);
        if constexpr (LPR <= 4) val += row_ror<4>(val);
        if constexpr (MODE == MODE_GIN) val *= scale;
        r[0] = val;
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float s = slot_reduce<LPR>(acc[k]);
            if constexpr (MODE == MODE_GIN) s *= scale;
            r[k] = s;
        }
    }
    return r;
}

// Checksum of 1024 evenly spaced entries of an id array (one wavefront; every lane returns the sum): what a packed copy
// of the ids remembers about the array it was made from, and what the prologue of a call compares again.
__device__ __forceinline__ unsigned long long sample_checksum(const int32_t *__restrict__ ids, int64_t n, int lane)
{
    unsigned long long sum = 0;
    if (n > 0) {
        uint32_t v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {            // all 16 loads in flight together
            const int i = k * 64 + lane;
            v[k] = (uint32_t)ids[(int64_t)(((unsigned long long)i * (unsigned long long)n) >> 10)];
        }
#pragma unroll
        for (int k = 0; k < 16; k++) sum += (unsigned long long)v[k] * (unsigned long long)(2 * (k * 64 + lane) + 1);
    }
    for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d);
    return sum;
}

// What a packed copy of the ids remembers about the graph it was made from: samples of column_index AND of part_pointers
// (the item starts of the copy are derived from the latter; its last entry -- the edge count -- is always among the
// samples' end points) -- 1024 + 1024 entries, so a buffer that was rewritten is noticed, a few changed entries are not.
__device__ __forceinline__ unsigned long long graph_checksum(const int32_t *__restrict__ ids, int64_t n,
                                                             const int32_t *__restrict__ pp, int64_t P, int lane)
{
    const unsigned long long a = sample_checksum(ids, n, lane);
    const unsigned long long b = sample_checksum(pp, P + 1, lane);
    const unsigned long long last = (unsigned long long)(uint32_t)pp[P];
    return a ^ ((b + last) * 0x9E3779B97F4A7C15ull);
}

}  // namespace gnna

#endif  // GNNA_DEVICE_H_
