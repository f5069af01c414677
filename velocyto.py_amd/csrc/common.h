// common.h -- shared helpers for the gfx950 kernels of libvelocyto_hip.so.
// CDNA4 only: wave64, 160 KiB LDS per CU, no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/velocyto_hip.h"

#define VCY_WAVE 64

namespace vcy {

extern thread_local char g_err[512];

inline int fail(int code, const char *fmt, const char *a = "", long long b = 0, long long c = 0)
{
    snprintf(g_err, sizeof(g_err), fmt, a, b, c);
    return code;
}

#define VCY_CHECK_HIP(expr)                                                                        \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) return vcy::fail(VCY_ERR_HIP, "%s (%lld) at line %lld", hipGetErrorString(_e), (long long)_e, __LINE__); \
    } while (0)

#define VCY_LAUNCH_CHECK() VCY_CHECK_HIP(hipGetLastError())

#define VCY_REQUIRE(cond, msg)                                                \
    do {                                                                      \
        if (!(cond)) return vcy::fail(VCY_ERR_INVALID, "%s", msg);            \
    } while (0)

// 16-byte vector of T
template <typename T> struct Vec;
template <> struct Vec<float> { using type = float4; static constexpr int N = 4; };
template <> struct Vec<double> { using type = double2; static constexpr int N = 2; };

// ---- wave64 reductions on DPP (no LDS crossbar traffic, unlike __shfl_xor -> ds_bpermute_b32).
// quad_perm xor1, xor2, row_half_mirror, row_mirror leave every lane of a 16-lane row with the row
// total; row_bcast15 / row_bcast31 (gfx9 family) chain the four rows; lane 63 then holds the wave
// total, which is broadcast through an SGPR.  Must be called with all 64 lanes active.
template <int CTRL, int ROWMASK> __device__ __forceinline__ int dpp_i32(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROWMASK, 0xF, false);   // lanes without a source read 0
}
template <int CTRL, int ROWMASK> __device__ __forceinline__ float dpp_val(float v) { return __int_as_float(dpp_i32<CTRL, ROWMASK>(__float_as_int(v))); }
template <int CTRL, int ROWMASK> __device__ __forceinline__ int dpp_val(int v) { return dpp_i32<CTRL, ROWMASK>(v); }
template <int CTRL, int ROWMASK> __device__ __forceinline__ unsigned dpp_val(unsigned v) { return (unsigned)dpp_i32<CTRL, ROWMASK>((int)v); }
template <int CTRL, int ROWMASK> __device__ __forceinline__ double dpp_val(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = dpp_i32<CTRL, ROWMASK>((int)(b & 0xffffffffll)), hi = dpp_i32<CTRL, ROWMASK>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ float lane63(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }
__device__ __forceinline__ int lane63(int v) { return __builtin_amdgcn_readlane(v, 63); }
__device__ __forceinline__ unsigned lane63(unsigned v) { return (unsigned)__builtin_amdgcn_readlane((int)v, 63); }
__device__ __forceinline__ double lane63(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

template <typename T> __device__ __forceinline__ T wave_sum(T v)
{
    v += dpp_val<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
    v += dpp_val<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
    v += dpp_val<0x141, 0xF>(v);   // row_half_mirror
    v += dpp_val<0x140, 0xF>(v);   // row_mirror
    v += dpp_val<0x142, 0xA>(v);   // row_bcast15 -> rows 1, 3
    v += dpp_val<0x143, 0xC>(v);   // row_bcast31 -> rows 2, 3
    return lane63(v);
}
template <typename T> __device__ __forceinline__ T wave_max(T v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        T o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

// block-wide sum through LDS scratch of (blockDim/64) entries; result valid in all threads.
template <typename T> __device__ __forceinline__ T block_sum(T v, T *scratch)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    T r = 0;
    for (int i = 0; i < nw; ++i) r += scratch[i];
    return r;
}

inline hipStream_t as_stream(vcy_stream s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace vcy
