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

// value of lane `l` (wave-uniform lane number) broadcast to the whole wave
__device__ __forceinline__ float readlane_t(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ double readlane_t(double v, int l)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
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
// ---- four wave totals for the price of one: transposing reduction on the gfx950 lane-swap instructions.
// v_permlane32_swap exchanges lanes 32..63 of its first operand with lanes 0..31 of its second, so after swap(a, c) the
// sum a + c holds, per lane column, the two-half partial of a in lanes 0..31 and that of c in lanes 32..63: one swap and
// one add halve two values at once.  v_permlane16_swap does the same with 16-lane rows.  Two levels leave row r of the
// result holding the 16 column partials of the r-th argument; four DPP adds inside the row finish it.  10 VALU
// instructions for four totals (3 swaps, 3 adds, 4 DPP adds) against 4 x (6 DPP adds + 2 moves + readlane) for four
// wave_sum calls.  Returns: every lane of row r (lanes 16r..16r+15) holds the wave total of argument r.
// Must be called with all 64 lanes active.
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void lane_swap32(float &a, float &b)
{
    const v2u_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r.x); b = __uint_as_float(r.y);
}
__device__ __forceinline__ void lane_swap16(float &a, float &b)
{
    const v2u_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r.x); b = __uint_as_float(r.y);
}
__device__ __forceinline__ void lane_swap32(double &a, double &b)
{
    const unsigned long long ba = (unsigned long long)__double_as_longlong(a), bb = (unsigned long long)__double_as_longlong(b);
    const v2u_t lo = __builtin_amdgcn_permlane32_swap((unsigned)ba, (unsigned)bb, false, false);
    const v2u_t hi = __builtin_amdgcn_permlane32_swap((unsigned)(ba >> 32), (unsigned)(bb >> 32), false, false);
    a = __longlong_as_double((long long)(((unsigned long long)hi.x << 32) | lo.x));
    b = __longlong_as_double((long long)(((unsigned long long)hi.y << 32) | lo.y));
}
__device__ __forceinline__ void lane_swap16(double &a, double &b)
{
    const unsigned long long ba = (unsigned long long)__double_as_longlong(a), bb = (unsigned long long)__double_as_longlong(b);
    const v2u_t lo = __builtin_amdgcn_permlane16_swap((unsigned)ba, (unsigned)bb, false, false);
    const v2u_t hi = __builtin_amdgcn_permlane16_swap((unsigned)(ba >> 32), (unsigned)(bb >> 32), false, false);
    a = __longlong_as_double((long long)(((unsigned long long)hi.x << 32) | lo.x));
    b = __longlong_as_double((long long)(((unsigned long long)hi.y << 32) | lo.y));
}
template <typename T> __device__ __forceinline__ T wave_sum_rows(T a, T b, T c, T d)
{
    lane_swap32(a, c);                 // a + c: lanes 0..31 <- a over both halves, lanes 32..63 <- c
    lane_swap32(b, d);
    T ac = a + c, bd = b + d;
    lane_swap16(ac, bd);               // ac + bd: row 0 <- a, row 1 <- b, row 2 <- c, row 3 <- d (16 column partials each)
    T v = ac + bd;
    v += dpp_val<0xB1, 0xF>(v);        // quad_perm [1,0,3,2]
    v += dpp_val<0x4E, 0xF>(v);        // quad_perm [2,3,0,1]
    v += dpp_val<0x141, 0xF>(v);       // row_half_mirror
    v += dpp_val<0x140, 0xF>(v);       // row_mirror: every lane of the row holds the row total
    return v;
}

// the same, stopped two steps early: every lane of a QUAD (4 lanes) of row r holds the sum of its quad's column partials of argument r;
// the four quads of a row add up to the wave total (the caller lets lanes 0, 4, 8, 12 of the row add theirs to one LDS word each)
template <typename T> __device__ __forceinline__ T wave_sum_rows_quads(T a, T b, T c, T d)
{
    lane_swap32(a, c);
    lane_swap32(b, d);
    T ac = a + c, bd = b + d;
    lane_swap16(ac, bd);
    T v = ac + bd;
    v += dpp_val<0xB1, 0xF>(v);        // quad_perm [1,0,3,2]
    v += dpp_val<0x4E, 0xF>(v);        // quad_perm [2,3,0,1]
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

// ---- process-wide facts, each computed once under a lock and then only read (layout.hip): nothing a call can observe
// changes between calls, so the entry points stay re-entrant from any number of host threads and devices.
struct DevInfo { int cus; int lds_optin; };                      // CU count, largest dynamic LDS a workgroup may ask for
int device_info(DevInfo *out);                                   // properties of the CURRENT device (cached per device id)
int ensure_dynamic_lds(const void *kernel, size_t bytes);        // hipFuncAttributeMaxDynamicSharedMemorySize, raised once per (device, kernel)
int env_int(const char *name, int dflt);                         // integer environment switch, read once per name

}  // namespace vcy
