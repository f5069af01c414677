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

template <typename T> __device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
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
