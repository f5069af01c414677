// coldeltacor.hip -- stage D: the cell x cell velocity-correlation kernels.
//
// Reference: the six x_colDeltaCor* kernels of velocyto/speedboosted.pyx:13-538.  For a cell
// c and another cell i they Pearson-correlate, over genes g,
//     A[g] = f(e[g,i] - e[g,c])      with      b[g] = d[g,c].
// The reference makes five streaming passes over a per-thread G x nrndm fp64 scratch and
// gathers e column-wise at stride C (a cache miss per element).  Here the matrices are
// cells-major (one cell = one contiguous gene vector) and every pair is ONE pass:
// raw moments  sum A, sum A^2, sum A*b  are accumulated while neighbour i's gene vector
// streams from HBM at 16 B/lane, and r = cov / sqrt(varA * varb) is formed at the end.
//
// HBM roofline: a (c, i) pair must read neighbour i's G elements; e[c], d[c] are read once
// per cell and parked in LDS.  Algorithmic bytes per cell = (nrndm + 2) * G * sizeof(T)
// + nrndm * (4 + sizeof(T)).  ~14 VALU lane-ops + 1 transcendental per 4 B loaded keeps the
// VALU at ~1/3 of its rate when HBM runs at 6 TB/s, so the one-cell-per-workgroup kernel
// (k_cdc_partial) is HBM-bound by design; the grouped kernel (k_cdc_partial_grouped) shares
// neighbour rows among 8 adjacent cells out of LDS, cuts the HBM-side traffic 10x and is
// VALU/transcendental-bound instead.
#include <stdlib.h>
#include "common.h"
#ifndef VCY_EXP
#define VCY_EXP 0
#endif
#include <type_traits>

namespace vcy {

template <typename T> __device__ __forceinline__ T fast_sqrt(T x);
template <> __device__ __forceinline__ float fast_sqrt<float>(float x) { return __builtin_amdgcn_sqrtf(x); }  // v_sqrt_f32, 1 ulp
template <> __device__ __forceinline__ double fast_sqrt<double>(double x) { return sqrt(x); }
template <typename T> __device__ __forceinline__ T fast_log10(T x);
template <> __device__ __forceinline__ float fast_log10<float>(float x) { return __builtin_amdgcn_logf(x) * 0.30102999566398120f; }  // v_log_f32 (log2)
template <> __device__ __forceinline__ double fast_log10<double>(double x) { return log10(x); }

// Element transform; branch rules per reference variant:
//   full    sqrt : t>0 ? sqrt(t+psc) : -sqrt(-t+psc)                 speedboosted.pyx:110-114
//   full    log10: t>0 ? log10(t+psc): -log10(-t+psc)                speedboosted.pyx:195-199
//   partial sqrt : |t|<1e-16 ? 0 : (t>0 ? sqrt(t+psc) : -sqrt(-t+psc))   speedboosted.pyx:372-378
//   partial log10: t>=0 ? log10(t+psc) : -log10(-t+psc)              speedboosted.pyx:469-473
template <typename T, int TR, int RULES> __device__ __forceinline__ T xform(T t, T psc)
{
    if (TR == VCY_LINEAR) return t;
    const T a = fabs(t) + psc;
    const T s = (TR == VCY_SQRT) ? fast_sqrt<T>(a) : fast_log10<T>(a);
    if (TR == VCY_SQRT && RULES != VCY_RULES_FULL)           // partial sqrt: one v_bfi (copysign) + one compare/select
        return (fabs(t) < T(1e-16)) ? T(0) : copysign(s, t);
    T r;
    if (TR == VCY_LOG10 && RULES == VCY_RULES_PARTIAL) r = (t >= T(0)) ? s : -s;
    else r = (t > T(0)) ? s : -s;
    return r;
}

// f32 fast path of the hot variant (partial sqrt).  Counters put the grouped kernel at ~95 % VALU-busy, so the only lever
// is issue slots per element; wave64 slots on gfx950: plain f32 op 1, v_sqrt_f32 2, v_pk_*_f32 2 (measured,
// tools/ubench/valu_rates.hip - packing buys nothing).  The literal rule costs add, sqrt, cmp, bfi, cndmask = 6 slots
// besides the subtract and the three accumulations; this form costs 5:
//     c = clamp(|t| * 2^54, 0, 1)      one v_mul with |src| and the clamp output modifier
//     u = fma(psc, c, |t|)             = |t| + psc (same rounding) when c == 1, 0 when t == 0
//     A = copysign(sqrt(u), t)         v_sqrt + v_bfi
// c is exactly 1 for |t| >= 2^-54 (5.6e-17) and exactly 0 for t == 0, so every element the rule of
// speedboosted.pyx:372-378 can tell apart in practice gets bit-identical values; only 0 < |t| < 1e-16 (unreachable for
// f32 differences of values above 1e-9) sees a ramp instead of a hard zero.  The f64 parity build keeps the literal rule.
template <> __device__ __forceinline__ float xform<float, VCY_SQRT, VCY_RULES_PARTIAL>(float t, float psc)
{
    // (measured in the kernel, round 2: `|t| + psc -> sqrt -> v_med3(s, -s, t * 2^100)` and psc held in a VGPR change nothing)
    const float c = __builtin_amdgcn_fmed3f(fabsf(t) * 0x1p54f, 0.0f, 1.0f);
    return copysignf(fast_sqrt<float>(fmaf(psc, c, fabsf(t))), t);
}

// f64 hot variant (partial sqrt; the reference-precision build).  sqrt(double) expands to v_rsq_f64 (12.5 clocks per wave64, measured:
// profiles/r03_valu_issue_f64.txt) + a coupled Goldschmidt step + two residual corrections, wrapped in a range scaling (inputs below
// 2^-767) and a 0 / inf fix-up: ~17 f64 instructions, 130 clocks per element with the moments.  The argument here is |t| + psc with
// |t| >= 1e-16 whenever the result is used, far inside the f32 exponent range, so the seed comes from the f32 unit instead:
//     x_f = (float) x,  y_f = v_rsq_f32(x_f) (8 clocks, 2^-23),  s0 = (double)(x_f * y_f): a 24-BIT seed of the root, 2^-22 off;
//     h   = y_f / 2 as a double: one conversion and an exponent decrement (an integer subtract on the high word, full rate);
//     d = x - s0^2 (exact: the square of a 24-bit number fits a double, the fma rounds once), s1 = s0 + d h   -> 2^-44;
//     d = x - s1^2 (one rounding),                                                       s  = s1 + d h   -> below half an ulp.
// Four f64 FMAs where round 3's Goldschmidt form (g = x y, r = 1/2 - h g, g += g r, one correction) took a multiply and four, and a
// result that is the correctly rounded root in all but near-tie cases (round 3: up to 2 ulp; tools/ubench/valu_issue_f64.hip counts
// both over 2^26 arguments in [2^-60, 2^60], profiles/r04_valu_issue_f64.txt).  Domain: |t| + psc in [1e-38, 3e38] - a count-derived
// matrix never leaves it, ops.check_f64_sqrt_domain refuses one that does, below 1e-16 the zero rule discards the value.
// The element of the hot loop: that seed and those two Newton corrections (tools/ubench/valu_issue_f64.hip: sqrt_f32prod), with the zero rule of speedboosted.pyx:372 applied to
// the ARGUMENT of the seed: v_rsq_f32(+inf) = +0, a y_f of 0 makes s0 = x_f y_f = 0 and h = y_f / 2 = 0 (halved in f32 here: an exponent
// decrement of the converted value would turn a zero into -inf), both corrections add (x - 0) 0 and the root comes out as an exact 0 -
// one v_cndmask on a 32-bit value where zeroing the finished double took two.  (The select sits before the v_rsq because the compiler
// moves one placed on y_f past the conversion to f64, where it is two again.)  18 instructions per element instead of 19: v_add_f64 (t),
// v_add_f64 (|t| + psc), v_cvt_f32_f64, v_cmp_f64 + v_cndmask (zero rule), v_rsq_f32, 2 x v_mul_f32 (s0, h), 2 x v_cvt_f64_f32, 4 x
// v_fma_f64, v_bfi (sign), v_add_f64 + 2 x v_fma_f64 (moments).  A discarded element is -0 where t < 0: it adds nothing to any moment.
// Measured on one box, stage D at 50k x 30k (profiles/r04b_elem_variants*.txt; boxes differ by +-4 %, forms compared within a call):
// 19-instruction form 245.2 / 241.2 ms, this one 232.6 / 232.4; s0 = x y and h = y / 2 as f64 products of the one converted seed (17
// instructions, two more on the f64 multiplier) 237.9 - slower than this although shorter: the launch clocks lower (GRBM_GUI_ACTIVE / time
// 2.27 -> 2.20 GHz), the f64 multiplier is what the power budget pays for; sum A^2 as sum |t| + psc x (kept elements, counted from the
// compare's mask on the scalar unit: a v_add_f64 for a v_fma_f64) 235.4 against 232.6 - no gain, the literal a^2 stays; h = rsq / 2 and
// s0 = 2 x_f h through output modifiers (17 instructions; needs the wave's IEEE bit and f32 denormals off, s_setreg at kernel start,
// and the four f32 instructions of two elements in one asm block for the trans-use hazard: parity-green) 238.9 against 242.4 on its
// box, 1.4 % - not kept; v_rsq_f64 on x itself with the rule on the high word of its result and both products in f64 (15 instructions, no
// conversions) 237.4 against 231.2 on its box.
template <> __device__ __forceinline__ double xform<double, VCY_SQRT, VCY_RULES_PARTIAL>(double t, double psc)
{
    const double x = fabs(t) + psc;
    const float xf0 = (float)x;
    const float xf = (fabs(t) < 1e-16) ? __builtin_inff() : xf0;
    const float yf = __builtin_amdgcn_rsqf(xf);
    const double s0 = (double)(xf0 * yf), h = (double)(0.5f * yf);
    double d = fma(-s0, s0, x);
    const double s1 = fma(d, h, s0);
    d = fma(-s1, s1, x);
    // (round 5: the sign as v_and_b32 + v_or_b32, 19 instructions: 241.0 against 235.0 ms; as one v_and_or_b32: 235.0 - it costs what v_bfi_b32 costs)
    return copysign(fma(d, h, s1), t);
}

// VCY_RULES_PARTIAL_NOPSC (f32, sqrt): A = sign(t) sqrt|t| as t * rsq|t| with the legacy multiply (0 * anything = 0: the zero
// rule of speedboosted.pyx:372 for free, the sign from t) - three instructions (v_sub, v_rsq_f32, v_mul_legacy_f32) where
// the literal rule needs five.  The pseudocount is dropped: in f32 `|t| + psc` IS `|t|` for every |t| >= 2^24 psc (1.7e-3 at
// the default 1e-10), below that the two rules differ by sqrt(|t| + psc) - sqrt|t| <= psc / (2 sqrt|t|).  The caller opts in
// (velocyto_hip.h; the Python layer does when psc <= 1e-9 and the matrix is of ordinary scale); the f64 build has no such
// form.  llvm.amdgcn.fmul.legacy has no clang builtin in ROCm 7.2: declared by its intrinsic name, so that the compiler
// (not an asm statement) places it and keeps the wait state a trans result needs before its first use.
extern "C" __device__ float vcy_fmul_legacy(float, float) __asm("llvm.amdgcn.fmul.legacy");
template <> __device__ __forceinline__ float xform<float, VCY_SQRT, VCY_RULES_PARTIAL_NOPSC>(float t, float)
{
    return vcy_fmul_legacy(t, __builtin_amdgcn_rsqf(fabsf(t)));
}

// Shifted moments.  Pearson's r does not change when a constant is subtracted from every A[g], so the log10 variants
// accumulate A[g] - K with K = f(0), the value of the transform where the two cells agree (log10(psc) = -10 for the
// default psc of 1e-10).  On count data most genes of a pair agree, so without the shift sum A^2 and (sum A)^2 / n are
// two numbers near 100 n whose difference - the variance - sits in the last digits of an f32 accumulator; with it the
// agreeing genes contribute exact zeros and the single-pass raw moments keep their accuracy in f32.  The full-rule sqrt
// variant has f(0) = -sqrt(psc) (t > 0 fails at 0): shifted too, so that identical cells give an exact zero variance
// (NaN, like the reference's centred sums) instead of f32 rounding noise.  Partial sqrt and linear have f(0) = 0.
template <int TR, int RULES> struct Shifted { static constexpr bool value = TR == VCY_LOG10 || (TR == VCY_SQRT && RULES == VCY_RULES_FULL); };   // f(0) != 0

template <typename T, int TR, int RULES> __device__ __forceinline__ T xform_shift(T psc)
{
    if (!Shifted<TR, RULES>::value) return T(0);
    const T k = xform<T, TR, RULES>(T(0), psc);
    return (k - k == T(0)) ? k : T(0);             // psc = 0 gives -inf: every agreeing gene is NaN in the reference as well
}
// f32 forms the logarithm as v_log_f32 (log2) times log10(2); the shift is taken in the log2 domain, BEFORE that scaling,
// so that an agreeing gene contributes an exact zero whatever the compiler contracts: (l - l0) * c, never fma(l, c, -K)
// (which leaves the rounding residue of l * c on every agreeing gene and turns a zero-variance self pair into noise).
template <> __device__ __forceinline__ float xform_shift<float, VCY_LOG10, VCY_RULES_PARTIAL>(float psc)
{
    const float l0 = __builtin_amdgcn_logf(psc);                // f(0) = +log10(psc): t >= 0 takes the positive branch
    return (l0 - l0 == 0.f) ? l0 : 0.f;
}
template <> __device__ __forceinline__ float xform_shift<float, VCY_LOG10, VCY_RULES_FULL>(float psc)
{
    const float l0 = -__builtin_amdgcn_logf(psc);               // f(0) = -log10(psc): t > 0 fails at t == 0
    return (l0 - l0 == 0.f) ? l0 : 0.f;
}

// A[g] - K: the transformed difference as it enters the moments
template <typename T, int TR, int RULES> __device__ __forceinline__ T xform_s(T t, T psc, T K)
{
    const T a = xform<T, TR, RULES>(t, psc);
    return Shifted<TR, RULES>::value ? a - K : a;
}
template <> __device__ __forceinline__ float xform_s<float, VCY_LOG10, VCY_RULES_PARTIAL>(float t, float psc, float K)
{
    const float l = __builtin_amdgcn_logf(fabsf(t) + psc);
    return (((t >= 0.f) ? l : -l) - K) * 0.30102999566398120f;
}
template <> __device__ __forceinline__ float xform_s<float, VCY_LOG10, VCY_RULES_FULL>(float t, float psc, float K)
{
    const float l = __builtin_amdgcn_logf(fabsf(t) + psc);
    return (((t > 0.f) ? l : -l) - K) * 0.30102999566398120f;
}

template <typename T> __device__ __forceinline__ T pearson_from_moments(double sA, double sAA, double sAb, double sb, double sbb, double n)
{
    const double cov = sAb - sA * sb / n;
    const double va = sAA - sA * sA / n;
    const double vb = sbb - sb * sb / n;
    return (T)(cov / sqrt(va * vb));  // va == 0 -> 0/0 = NaN, like the reference's 0 * inf
}

// Optional fusion of the velocity chain into the staging of d[c] (stage C folded into stage D): instead of reading
// a materialised dmat row, the group's members compute it on the fly from Ux, Sx (= e), gamma and q,
//     dmat = sign(D) f(|D| + psc),  D = (Sx + used_dt * dt_shift * (Ux - (gamma Sx + q))) - Sx
// (analysis.py:1346, 1369, 1399, 1538, 1575-1601; identical arithmetic to k_velocity_chain), which removes one
// 6 GB write + read per pass at 50k x 30k.  Ux == nullptr: d is read as given.
template <typename T> struct FuseArgs {
    const T *Ux;
    const float *gamma, *q;
    T dt_shift, used_dt;
};

template <typename T, int TR> __device__ __forceinline__ T fused_dmat(T s, T u, float gm, float qq, T dt_shift, T used_dt, T psc)
{
    const T upred = (T)gm * s + (T)qq;
    const T ds = dt_shift * (u - upred);
    const T D = (s + used_dt * ds) - s;
    if (TR == VCY_LINEAR) return D;
    const T a = fabs(D) + psc;
    const T f = (TR == VCY_SQRT) ? fast_sqrt<T>(a) : fast_log10<T>(a);
    return D > T(0) ? f : (D < T(0) ? -f : T(0) * f);
}

// ---------------------------------------------------------------------------------------------
// Partial kernel: one workgroup per cell c.  Genes are walked in chunks of `gchunk`; the chunk of
// e[c] and d[c] is staged in LDS (2 * gchunk * sizeof(T) bytes), then each WAVE takes neighbours
// n = wave, wave + nwaves, ... and streams row e[ixs[c,n]] for that chunk with 16-byte loads,
// UNROLL loads in flight per lane.  Cross-chunk partial moments live in LDS (acc[3*nrndm]); only
// the owning wave touches acc[n], so there are no atomics and the result is deterministic.
template <typename T, int TR, int RULES>
__global__ __launch_bounds__(1024) void k_cdc_partial(const T *__restrict__ e, const T *__restrict__ d,
                                                       const int32_t *__restrict__ ixs, T *__restrict__ out,
                                                       const int32_t *__restrict__ order, int G, int64_t ld,
                                                       int64_t cell0, int64_t d_row0, int C_out, int nrndm, int gchunk, T psc, FuseArgs<T> fuse)
{
    using V = typename Vec<T>::type;
    constexpr int N = Vec<T>::N;
    constexpr int UNROLL = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T *ec = reinterpret_cast<T *>(smem);
    T *dc = ec + gchunk;
    T *acc = dc + gchunk;                                       // [3 * nrndm]
    double *red = reinterpret_cast<double *>(acc + 3 * ((nrndm + 1) & ~1));  // [32] block-reduce scratch (8-byte aligned)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int per_x = (C_out + 7) / 8;                            // XCD-aware schedule (see k_cdc_partial_grouped)
    const int pos = ((int)blockIdx.x & 7) * per_x + ((int)blockIdx.x >> 3);
    if (pos >= C_out) return;
    const int cl = order ? order[pos] : pos;                      // local output row
    const int64_t c = cell0 + cl;
    const T *erow_c = e + c * ld;
    const T *drow_c = (fuse.Ux ? fuse.Ux : d) + (c - d_row0) * ld;          // fused: the cell's Ux row, d[c] is built while staging

    for (int n = tid; n < 3 * nrndm; n += blockDim.x) acc[n] = T(0);
    double sb = 0.0, sbb = 0.0;
    const T K = xform_shift<T, TR, RULES>(psc);

    for (int g0 = 0; g0 < G; g0 += gchunk) {
        const int gl = min(gchunk, G - g0);
        const int nvec = gl / N;
        __syncthreads();  // previous chunk fully consumed (also orders the acc zeroing)
        for (int v = tid; v < nvec; v += blockDim.x) {
            const V ev = reinterpret_cast<const V *>(erow_c + g0)[v];
            V dv = reinterpret_cast<const V *>(drow_c + g0)[v];
            if (fuse.Ux) {
                const T *ep = reinterpret_cast<const T *>(&ev);
                T *dq = reinterpret_cast<T *>(&dv);
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    const int g = g0 + v * N + k;
                    dq[k] = fused_dmat<T, TR>(ep[k], dq[k], fuse.gamma[g], fuse.q ? fuse.q[g] : 0.f, fuse.dt_shift, fuse.used_dt, psc);
                }
            }
            reinterpret_cast<V *>(ec)[v] = ev;
            reinterpret_cast<V *>(dc)[v] = dv;
            const T *dp = reinterpret_cast<const T *>(&dv);
#pragma unroll
            for (int k = 0; k < N; ++k) { sb += (double)dp[k]; sbb += (double)dp[k] * (double)dp[k]; }
        }
        for (int g = nvec * N + tid; g < gl; g += blockDim.x) {  // < N scalar tail elements
            const T ev = erow_c[g0 + g];
            T dv = drow_c[g0 + g];
            if (fuse.Ux) dv = fused_dmat<T, TR>(ev, dv, fuse.gamma[g0 + g], fuse.q ? fuse.q[g0 + g] : 0.f, fuse.dt_shift, fuse.used_dt, psc);
            ec[g] = ev; dc[g] = dv;
            sb += (double)dv; sbb += (double)dv * (double)dv;
        }
        __syncthreads();

        for (int n = wave; n < nrndm; n += nwaves) {
            const int i = __builtin_amdgcn_readfirstlane(ixs[(int64_t)cl * nrndm + n]);
            const T *row = e + (int64_t)i * ld + g0;
            T sA[N], sAA[N], sAb[N];
#pragma unroll
            for (int k = 0; k < N; ++k) { sA[k] = T(0); sAA[k] = T(0); sAb[k] = T(0); }
            int v = lane;
            for (; v + 64 * (UNROLL - 1) < nvec; v += 64 * UNROLL) {
                V x[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) x[u] = reinterpret_cast<const V *>(row)[v + 64 * u];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const V ecv = reinterpret_cast<const V *>(ec)[v + 64 * u];
                    const V dcv = reinterpret_cast<const V *>(dc)[v + 64 * u];
                    const T *xp = reinterpret_cast<const T *>(&x[u]);
                    const T *ep = reinterpret_cast<const T *>(&ecv);
                    const T *bp = reinterpret_cast<const T *>(&dcv);
#pragma unroll
                    for (int k = 0; k < N; ++k) {
                        const T tt = xp[k] - ep[k];
                        T a = xform_s<T, TR, RULES>(tt, psc, K);
                        sA[k] += a;
                        if (RULES == VCY_RULES_PARTIAL_NOPSC) sAA[k] += fabs(tt);      // A^2 = |t| exactly (as in the grouped kernel)
                        else sAA[k] = fma(a, a, sAA[k]);
                        sAb[k] = fma(a, bp[k], sAb[k]);
                    }
                }
            }
            for (; v < nvec; v += 64) {
                const V xv = reinterpret_cast<const V *>(row)[v];
                const V ecv = reinterpret_cast<const V *>(ec)[v];
                const V dcv = reinterpret_cast<const V *>(dc)[v];
                const T *xp = reinterpret_cast<const T *>(&xv);
                const T *ep = reinterpret_cast<const T *>(&ecv);
                const T *bp = reinterpret_cast<const T *>(&dcv);
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    const T tt = xp[k] - ep[k];
                    T a = xform_s<T, TR, RULES>(tt, psc, K);
                    sA[k] += a;
                    if (RULES == VCY_RULES_PARTIAL_NOPSC) sAA[k] += fabs(tt);
                    else sAA[k] = fma(a, a, sAA[k]);
                    sAb[k] = fma(a, bp[k], sAb[k]);
                }
            }
            {
                const int g = nvec * N + lane;
                if (g < gl) {
                    const T tt = row[g] - ec[g];
                    T a = xform_s<T, TR, RULES>(tt, psc, K);
                    sA[0] += a;
                    if (RULES == VCY_RULES_PARTIAL_NOPSC) sAA[0] += fabs(tt);
                    else sAA[0] = fma(a, a, sAA[0]);
                    sAb[0] = fma(a, dc[g], sAb[0]);
                }
            }
            T tA = sA[0], tAA = sAA[0], tAb = sAb[0];
#pragma unroll
            for (int k = 1; k < N; ++k) { tA += sA[k]; tAA += sAA[k]; tAb += sAb[k]; }
            tA = wave_sum(tA); tAA = wave_sum(tAA); tAb = wave_sum(tAb);
            if (lane == 0) {
                acc[3 * n + 0] += tA;
                acc[3 * n + 1] += tAA;
                acc[3 * n + 2] += tAb;
            }
        }
    }
    sb = block_sum(sb, red);
    sbb = block_sum(sbb, red);   // block_sum's leading __syncthreads also publishes acc[]
    for (int n = tid; n < nrndm; n += blockDim.x)
        out[(int64_t)cl * nrndm + n] = pearson_from_moments<T>((double)acc[3 * n], (double)acc[3 * n + 1],
                                                               (double)acc[3 * n + 2], sb, sbb, (double)G);
}

// ---------------------------------------------------------------------------------------------
// Grouped partial kernel: the same numbers as k_cdc_partial, but a workgroup owns GC cells that are
// adjacent in the schedule order (Morton order of the embedding => near each other) and walks the
// UNION of their neighbour lists, so a neighbour row shared by several cells of the group is read
// from HBM once and correlated against each of them out of LDS.  At 50k cells / nrndm 250 a group
// of 8 shares each row 3.5x on average (tools/neighbor_overlap.py), which moves the kernel from the
// HBM roofline to the VALU/transcendental one.
//   1. (neighbour, member, slot) keys of the group are bitonic-sorted in LDS; a run of equal neighbours is a ROW and
//      gets one 8-byte descriptor: neighbour << 19 | mask of the members that list it << 11 | first pair of the run.
//   2. genes are walked in chunks of NV*64 vectors; e[c_m], d[c_m] of all members are staged in LDS.
//   3. waves draw rows four at a time from an LDS counter: a wave loads the row chunk ONCE into registers (the next row
//      is in flight meanwhile), then for each member of the mask accumulates the three raw moments against that member's
//      LDS copy - operands read one vector ahead of the arithmetic -, reduces over the wave and adds into acc[pair]
//      (ds_add_f32 by the only wave that owns the pair in this chunk: program order, deterministic).
//      No LDS round trip is left exposed in the row loop ("the latency chain removed").
// Dual control (DUAL): estimate_transition_prob's default computes every correlation twice, against d and against the
// randomised control d2 = f(permute_rows_nsign(delta_S)) (analysis.py:1539-1542, 1578-1601).  Both passes share e, the
// neighbour lists and therefore every A = f(e_i - e_c); the dual kernel stages d2[c] beside d[c] and keeps a fourth
// running moment sum A*b2 per pair: one more FMA and one more LDS read per element instead of a second launch.
// Shape of a workgroup: GC cells, chunks of NV 16-byte vectors per lane (NV * 64 * 4 f32 genes).  The dual variant stages
// three arrays per member in the same 160 KiB: 6 cells x 1536 genes (the d-moment partials share the memory of the row
// descriptors to make it fit).  Same chunk length as the single kernel, hence the same order of summation: each dual output
// equals the single launch bit for bit.  Measured at 50k x 30k, nrndm 250, literal rule: 6 x 1536 103.2 ms, 8 x 1024 105.3 ms,
// one single-control launch 90.6 ms (before the row-loop restructure 6 x 1536 lost to 8 x 1024, 118.5 vs 110.0 ms).
constexpr int GRP_NV = 6;
constexpr int GRP_GC_DUAL = 6, GRP_NV_DUAL = 6;                 // dual control: three staged arrays per member -> 6 cells at the single kernel's chunk length
constexpr int GRP_GC = 8;
constexpr int GRP_GC_F64 = 6, GRP_NV_F64 = 8;                   // f64, single control: 6 cells, chunks of 8 vectors per lane (1024 genes)
constexpr int GRP_GC_DUAL_F64 = 4;                              // f64, dual control: 4 cells at the same chunk length

// dynamic LDS of one workgroup: staged rows, sort keys, per-pair accumulators, segment heads, scalars (also used by the host)
template <typename T> __host__ __device__ inline size_t grouped_lds_bytes(int gc, int nv, bool dual, int64_t maxpairs, int npad)
{
    const int as = dual ? 4 : 3;
    return (size_t)(dual ? 3 : 2) * gc * nv * 64 * 16 + (size_t)npad * 8 + sizeof(T) * (size_t)as * ((maxpairs + 1) & ~(int64_t)1) +
           8 * (size_t)(maxpairs + 2 > 64 + 4 * gc ? maxpairs + 2 : 64 + 4 * gc) + (size_t)(gc + 18) * sizeof(int) + 16;
}

template <typename T, int TR, int RULES, int GC, int NV, bool DUAL>
__global__ __launch_bounds__(1024) void k_cdc_partial_grouped(const T *__restrict__ e, const T *__restrict__ d, const T *__restrict__ d2,
                                                               const int32_t *__restrict__ ixs, T *__restrict__ out, T *__restrict__ out2,
                                                               const int32_t *__restrict__ order, int G, int64_t ld, int64_t cell0,
                                                               int64_t d_row0, int C_main, int tile_main, int C_tail, int tile_tail, int nrndm_all, int stride, int npad, T psc,
                                                               FuseArgs<T> fuse)
{
    using V = typename Vec<T>::type;
    constexpr int N = Vec<T>::N;
    constexpr int GCHUNK = NV * 64 * N;                         // genes per chunk
    constexpr int AS = DUAL ? 4 : 3;                            // running moments per pair: sum A, sum A^2, sum A b [, sum A b2]
    constexpr int PW = DUAL ? 4 : 2;                            // d-moment partials per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int maxpairs = GC * max(tile_main, tile_tail);        // LDS layout is sized for the widest tile
    T *ec = reinterpret_cast<T *>(smem);                        // [GC][GCHUNK]
    T *dc = ec + GC * GCHUNK;                                   // [GC][GCHUNK]
    T *dc2 = dc + GC * GCHUNK;                                  // [GC][GCHUNK] (DUAL only)
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(dc + (DUAL ? 2 : 1) * GC * GCHUNK);   // [npad]
    T *acc = reinterpret_cast<T *>(keys + npad);                // [AS * maxpairs]
    unsigned long long *desc = reinterpret_cast<unsigned long long *>(acc + AS * ((maxpairs + 1) & ~1));   // [maxpairs + 2] row descriptors
    // the d-moment partials ([64] per wave, then [GC] x 4 totals) are written after the last chunk, when the row descriptors
    // are dead: they share the descriptors' memory (grouped_lds_bytes sizes it for the larger of the two)
    double *part = reinterpret_cast<double *>(desc);
    // (no static __shared__: statics would precede the dynamic region and break its 16-byte alignment)
    double *s_sb = part + 64, *s_sbb = s_sb + GC, *s_sb2 = s_sbb + GC, *s_sbb2 = s_sb2 + GC;   // [GC] each
    int *s_cells = reinterpret_cast<int *>(desc + max(maxpairs + 2, 64 + 4 * GC));        // [GC]
    int *s_wavetot = s_cells + GC;                              // [16]
    int &s_U = s_wavetot[16];
    int *s_next = s_wavetot + 17;                               // dynamic row counter

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    // XCD-aware schedule: workgroup b runs on XCD b % 8 (observed; speed only) -> XCD x owns the contiguous
    // range of groups [x*per, (x+1)*per), so groups that share neighbour rows share one L2
    // One launch, two parts.  Main part: the first C_main cells of the schedule, neighbour lists (nrndm_all columns, row
    // pitch `stride`) in column tiles of tile_main.  Tail part: the C_tail cells after them - the groups beyond the last
    // full round of the device - in narrower tiles of tile_tail, so that the blocks dispatched last are short ones.
    // block -> (group, tile); the tiles of one group are gblocks apart.
    const int nb_main = ((((C_main + GC - 1) / GC) + 7) / 8 * 8) * (C_main > 0 ? (nrndm_all + tile_main - 1) / tile_main : 0);
    const bool tail = (int)blockIdx.x >= nb_main;
    const int bx = tail ? (int)blockIdx.x - nb_main : (int)blockIdx.x;
    const int C_out = tail ? C_tail : C_main, pos0 = tail ? C_main : 0, tilew = tail ? tile_tail : tile_main;
    const int ngroups = (C_out + GC - 1) / GC, per = (ngroups + 7) / 8, gblocks = per * 8;
    const int tile = bx / gblocks, bg = bx - tile * gblocks;
    const int gpos = (bg & 7) * per + (bg >> 3);
    if (gpos >= ngroups) return;
    const int n0 = tile * tilew;
    const int nrndm = min(tilew, nrndm_all - n0);
    ixs += n0;
    out += n0;
    if (DUAL) out2 += n0;
    const int g0cell = gpos * GC;
    const int gcount = min(GC, C_out - g0cell);
    const int npairs = gcount * nrndm;
    if (tid < GC) s_cells[tid] = tid < gcount ? (order ? order[pos0 + g0cell + tid] : pos0 + g0cell + tid) : 0;
    __syncthreads();
    // ---- 1. keys = (neighbour << 16) | (member << 12) | slot, sorted
    for (int t = tid; t < npad; t += blockDim.x) {
        unsigned long long key = ~0ull;
        if (t < npairs) {
            const int m = t / nrndm, n = t - m * nrndm;
            const unsigned i = (unsigned)ixs[(int64_t)s_cells[m] * stride + n];        // stride: row pitch of ixs / out (a tile of a wider list)
            key = ((unsigned long long)i << 16) | ((unsigned long long)m << 12) | (unsigned)n;
        }
        keys[t] = key;
    }
    __syncthreads();
    for (int size = 2; size <= npad; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < npad / 2; t += blockDim.x) {
                const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const unsigned long long a = keys[lo], b = keys[hi];
                if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
            }
            __syncthreads();
        }
    }
    // rows: a run of pairs with the same neighbour (its members ascend; a member listed twice for one neighbour starts a
    // new run).  desc[r] = neighbour << 19 | member mask << 11 | first pair: everything a wave needs to fetch the row and to
    // walk its pairs comes from ONE LDS word (block-wide exclusive scan: each thread owns a contiguous run of keys)
    {
        static_assert(GC <= 8, "member mask is 8 bits");
        auto head = [&](int t) {
            if (t == 0) return true;
            const unsigned long long a = keys[t] >> 12, b = keys[t - 1] >> 12;
            return (a >> 4) != (b >> 4) || a == b;
        };
        const int per = (npad + blockDim.x - 1) / blockDim.x;
        const int t0 = tid * per, t1 = min(npairs, t0 + per);
        int cnt = 0;
        for (int t = t0; t < t1; ++t) cnt += head(t) ? 1 : 0;
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
        if (lane == 63) s_wavetot[wave] = incl;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wave; ++w) base += s_wavetot[w];
        int rank = base + incl - cnt;
        for (int t = t0; t < t1; ++t)
            if (head(t)) {
                unsigned mask = 0;
                int q = t;
                do { mask |= 1u << (unsigned)((keys[q] >> 12) & 15); ++q; } while (q < npairs && !head(q));
                desc[rank++] = ((keys[t] >> 16) << 19) | ((unsigned long long)mask << 11) | (unsigned)t;
            }
        if (tid == blockDim.x - 1) { s_U = base + incl; }
        __syncthreads();
    }
    for (int t = tid; t < AS * npairs; t += blockDim.x) acc[t] = T(0);
    __syncthreads();
    const int U = s_U;
    const T K = xform_shift<T, TR, RULES>(psc);
    // staging roles: wave w stages member (w % GC), interleaved with the other waves of that member
    const int sm = wave % GC, sh = wave / GC, snh = max(1, nwaves / GC);
    double psb = 0.0, psbb = 0.0, psb2 = 0.0, psbb2 = 0.0;

    for (int g0 = 0; g0 < G; g0 += GCHUNK) {
        const int gl = min(GCHUNK, G - g0);
        const int nvec = (gl + N - 1) / N;                       // last vector may be partial: rows are zero-padded to ld
        const bool ragged = (gl % N) != 0;
        __syncthreads();
        if (tid == 0) *s_next = 0;
        if (sh < snh && sm < gcount) {
            const int64_t c = cell0 + s_cells[sm];
            const T *er = e + c * ld + g0;
            const T *dr = (fuse.Ux ? fuse.Ux : d) + (c - d_row0) * ld + g0;     // fused: the member's Ux row
            const T *dr2 = DUAL ? d2 + (c - d_row0) * ld + g0 : nullptr;
            for (int v = sh * 64 + lane; v < nvec; v += 64 * snh) {
                V ev = reinterpret_cast<const V *>(er)[v];
                V dv = reinterpret_cast<const V *>(dr)[v];
                V dv2;
                if (DUAL) dv2 = reinterpret_cast<const V *>(dr2)[v];
                T *dp = reinterpret_cast<T *>(&dv);
                T *dp2 = reinterpret_cast<T *>(&dv2);
                T *ep = reinterpret_cast<T *>(&ev);
                if (fuse.Ux) {
#pragma unroll
                    for (int k = 0; k < N; ++k) {
                        const int g = g0 + v * N + k;
                        if (g < G) dp[k] = fused_dmat<T, TR>(ep[k], dp[k], fuse.gamma[g], fuse.q ? fuse.q[g] : 0.f, fuse.dt_shift, fuse.used_dt, psc);
                    }
                }
                if (ragged && v == nvec - 1) {
#pragma unroll
                    for (int k = 0; k < N; ++k) if (k >= gl - v * N) { dp[k] = T(0); ep[k] = T(0); if (DUAL) dp2[k] = T(0); }
                }
                reinterpret_cast<V *>(ec + sm * GCHUNK)[v] = ev;
                reinterpret_cast<V *>(dc + sm * GCHUNK)[v] = dv;
                if (DUAL) reinterpret_cast<V *>(dc2 + sm * GCHUNK)[v] = dv2;
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    psb += (double)dp[k]; psbb += (double)dp[k] * (double)dp[k];
                    if (DUAL) { psb2 += (double)dp2[k]; psbb2 += (double)dp2[k] * (double)dp2[k]; }
                }
            }
        }
        if (gl < GCHUNK && sh < snh && sm < gcount)             // short last chunk: the lanes beyond it see zeros (f(0 - 0) adds nothing)
            for (int v = nvec + sh * 64 + lane; v < NV * 64; v += 64 * snh) {
                reinterpret_cast<V *>(ec + sm * GCHUNK)[v] = V{};
                reinterpret_cast<V *>(dc + sm * GCHUNK)[v] = V{};
                if (DUAL) reinterpret_cast<V *>(dc2 + sm * GCHUNK)[v] = V{};
            }
        __syncthreads();
        // ---- rows of this wave.  Nothing in here waits for a latency it could have started earlier:
        //  * the next row's chunk is in flight (registers xb) while the pairs of the current row (xa) are evaluated;
        //  * rows are drawn four at a time from an LDS counter (dynamic: rows carry 1..GC pairs, a static split leaves waves
        //    idle at the chunk barrier); the ticket for the next four is requested at the start of a quad and read two rows
        //    later, their descriptors are requested then and read one row later;
        //  * the operand vectors (ec, dc[, dc2] of the pair's member) are read one vector ahead of the arithmetic, across
        //    pair boundaries too (the first vector of the next pair before this pair's reduction); two register buffers
        //    alternate (NV is even: no copies at the loop edge).
        static_assert(NV % 2 == 0, "operand buffers alternate");
#define VCY_FENCE() __builtin_amdgcn_sched_barrier(0)
        // (a full chunk - every chunk but the last - loads its rows without predicates: the guarded form costs a v_cmp + s_and_saveexec +
        //  branch per vector, and in the f64 dual kernel the eight lane offsets it keeps for the compares were spilled and each reload
        //  put an s_waitcnt vmcnt(0) in front of the next row load; the two forms are two instances of the row loop below)
        auto load_row = [&](auto fullc, V (&x)[NV], unsigned long long dsc) {
            const T *row = e + (int64_t)(dsc >> 19) * ld + g0;
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int v = lane + 64 * u;
                if (decltype(fullc)::value || v < nvec) x[u] = reinterpret_cast<const V *>(row)[v];
                else x[u] = V{};                                 // short last chunk: zeros against the zeros staged below
            }
        };
        auto eval_row = [&](const V (&x)[NV], unsigned long long dsc) {
            int p = (int)(dsc & 2047);
            unsigned mask = (unsigned)(dsc >> 11) & 255u;
            int m = __builtin_ctz(mask);
            constexpr int RD = 2;       // operand buffers: RD - 1 vectors are in flight ahead of the arithmetic (3 buffers: 77.2 vs 76.8 ms, no gain)
            static_assert(NV % RD == 0, "operand buffers rotate");
            V eR[RD], bR[RD], b2R[RD];
            auto rd = [&](V &ev, V &bv, V &b2v, int mm, int u) {
                ev = reinterpret_cast<const V *>(ec + mm * GCHUNK)[lane + 64 * u];
                bv = reinterpret_cast<const V *>(dc + mm * GCHUNK)[lane + 64 * u];
                if (DUAL) b2v = reinterpret_cast<const V *>(dc2 + mm * GCHUNK)[lane + 64 * u];
            };
#pragma unroll
            for (int u = 0; u < RD - 1; ++u) rd(eR[u], bR[u], b2R[u], m, u);
            while (mask) {
                mask &= mask - 1;
                const int mn = mask ? __builtin_ctz(mask) : m;
                T sA, sAA, sAb, sAb2;                       // ONE partial sum per moment and lane: an element is 18 instructions, so the next add to an
                                                            // accumulator is issued long after the last one landed; two interleaved partials cost three adds
                                                            // per pair-chunk and six (f64: twelve) registers - f64 227.5 -> 226.2 ms, f64 dual 247.2 -> 236.5
                                                            // (fewer spills), f32 70.8 -> 70.4
                // (no zeroing: the first element of a pair initialises the sums; sum A^2 of the no-pseudocount rule is sum |t| - A^2 = |t|
                //  exactly for A = sign(t) sqrt|t|, one v_add with the |.| modifier that does not wait for the v_rsq_f32)
                auto fold = [&](const V &xv, const V &ev, const V &bv, const V &b2v, bool first) {
                    const T *xp = reinterpret_cast<const T *>(&xv);
                    const T *ep = reinterpret_cast<const T *>(&ev);
                    const T *bp = reinterpret_cast<const T *>(&bv);
                    const T *bp2 = reinterpret_cast<const T *>(&b2v);
                    constexpr bool ABS2 = RULES == VCY_RULES_PARTIAL_NOPSC;     // sum A^2 taken as sum |t|
#pragma unroll
                    for (int k = 0; k < N; ++k) {
                        const T tt = xp[k] - ep[k];
                        const T a = xform_s<T, TR, RULES>(tt, psc, K);
                        if (first && k == 0) {
                            sA = a;
                            sAA = ABS2 ? fabs(tt) : a * a;
                            sAb = a * bp[k];
                            if (DUAL) sAb2 = a * bp2[k];
                            continue;
                        }
                        sA += a;
                        if (ABS2) sAA += fabs(tt);
                        else sAA = fma(a, a, sAA);
                        sAb = fma(a, bp[k], sAb);
                        if (DUAL) sAb2 = fma(a, bp2[k], sAb2);
                    }
                };
#pragma unroll
                for (int u = 0; u < NV; ++u) {                  // vector u sits in buffer u % RD; the read of vector u + RD - 1 (of the next
                    constexpr int AHEAD = RD - 1;                //  pair past the end of this one) is issued before vector u is folded
                    const int un = u + AHEAD;
                    if (un < NV) rd(eR[un % RD], bR[un % RD], b2R[un % RD], m, un);
                    else rd(eR[un % RD], bR[un % RD], b2R[un % RD], mn, un - NV);
                    VCY_FENCE();
                    fold(x[u], eR[u % RD], bR[u % RD], b2R[u % RD], u == 0);
                    VCY_FENCE();
                }
                // the three (dual: four) wave totals in one transposing reduction: row r of `tot` holds moment r (the single-control
                // kernel feeds a fourth value nobody reads, so that both variants sum in the same order);
                // the last two steps of the row reduction are left to the LDS: the four quad totals of a row go into the pair's word as
                // four lanes of ONE ds_add (same address: the LDS adds them one after the other, in lane order).  Without return: the
                // wave that owns the pair is the only writer of acc[p][.], so the order of the additions is its program order
                // (deterministic) and nothing waits for the old value.  (Same box, stage D at 50k x 30k: f64 235.6 -> 232.0 ms, f32 71.9 -> 71.2;
                //  one step fewer still - pairs, eight lanes per row - f64 224.6 vs 225.2 but f32 75.8 vs 70.7; none at all f64 226.6, f32 109.9.)
                const T tot = wave_sum_rows_quads(sA, sAA, sAb, DUAL ? sAb2 : T(0));
                if ((lane & 3) == 0 && (lane >> 4) < AS)
                    __hip_atomic_fetch_add(&acc[AS * p + (lane >> 4)], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                ++p;
                m = mn;
            }
        };
        // tickets: quads of rows first, then - for the last TAIL rows of the chunk - pairs: the waves reach the chunk barrier within one
        // ticket of each other, and a pair is half the wait of a quad (round 5, one box, stage D at 50k x 30k: f64 234.7 -> 233.8 ms,
        // f32 72.7 -> 71.7; profiles/r05_tail_tickets.txt)
        auto rows = [&](auto fullc) {
            auto ticket = [&]() { int t = 0; if (lane == 0) t = atomicAdd(s_next, 1); return t; };     // lane 0 holds the value
            auto uni = [&](unsigned long long v) {
                const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
                const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
                return ((unsigned long long)hi << 32) | lo;
            };
            constexpr int TAIL = 64;
            const int U4 = U > TAIL ? ((U - TAIL) & ~3) : 0, T4 = U4 >> 2;
            auto first_row = [&](int v) { return v < T4 ? 4 * v : U4 + 2 * (v - T4); };
            V xa[NV], xb[NV];
            int v = __builtin_amdgcn_readfirstlane(ticket());
            int q = min(first_row(v), U);
            unsigned long long w0 = 0, w1 = 0, w2 = 0, w3 = 0;
            if (q < U) { w0 = uni(desc[q]); w1 = uni(desc[min(q + 1, U - 1)]); w2 = uni(desc[min(q + 2, U - 1)]); w3 = uni(desc[min(q + 3, U - 1)]); }
            if (q < U) load_row(fullc, xa, w0);
            while (v < T4) {                                     // a quad: rows q .. q + 3, all below U4
                const int tv = ticket();
                load_row(fullc, xb, w1);
                eval_row(xa, w0);
                load_row(fullc, xa, w2);
                eval_row(xb, w1);
                const int vn = __builtin_amdgcn_readfirstlane(tv);
                const int qn = min(first_row(vn), U);
                unsigned long long n0 = 0, n1 = 0, n2 = 0, n3 = 0;
                if (qn < U) { n0 = desc[qn]; n1 = desc[min(qn + 1, U - 1)]; n2 = desc[min(qn + 2, U - 1)]; n3 = desc[min(qn + 3, U - 1)]; }
                load_row(fullc, xb, w3);
                eval_row(xa, w2);
                if (qn < U) { n0 = uni(n0); n1 = uni(n1); n2 = uni(n2); n3 = uni(n3); }
                if (qn < U) load_row(fullc, xa, n0);
                eval_row(xb, w3);
                v = vn; q = qn; w0 = n0; w1 = n1; w2 = n2; w3 = n3;
            }
            while (q < U) {                                      // a pair: rows q, q + 1
                const int tv = ticket();
                const bool two = q + 1 < U;
                if (two) load_row(fullc, xb, w1);
                eval_row(xa, w0);
                const int vn = __builtin_amdgcn_readfirstlane(tv);
                const int qn = min(first_row(vn), U);
                unsigned long long n0 = 0, n1 = 0;
                if (qn < U) { n0 = uni(desc[qn]); n1 = uni(desc[min(qn + 1, U - 1)]); }
                if (qn < U) load_row(fullc, xa, n0);
                if (two) eval_row(xb, w1);
                v = vn; q = qn; w0 = n0; w1 = n1;
            }
        };
        if (nvec == 64 * NV) rows(std::true_type{}); else rows(std::false_type{});
    }
    psb = wave_sum(psb); psbb = wave_sum(psbb);
    if (DUAL) { psb2 = wave_sum(psb2); psbb2 = wave_sum(psbb2); }
    __syncthreads();                                            // every wave is through with the row descriptors: `part` takes their place
    if (lane == 0) {
        part[PW * wave] = psb; part[PW * wave + 1] = psbb;
        if (DUAL) { part[PW * wave + 2] = psb2; part[PW * wave + 3] = psbb2; }
    }
    __syncthreads();
    if (tid < gcount) {
        double a = 0.0, b = 0.0, a2 = 0.0, b2 = 0.0;
        for (int h = 0; h < snh; ++h) {
            const int w = tid + h * GC;
            if (w < nwaves) { a += part[PW * w]; b += part[PW * w + 1]; if (DUAL) { a2 += part[PW * w + 2]; b2 += part[PW * w + 3]; } }
        }
        s_sb[tid] = a; s_sbb[tid] = b; s_sb2[tid] = a2; s_sbb2[tid] = b2;
    }
    __syncthreads();
    for (int p = tid; p < npairs; p += blockDim.x) {
        const unsigned long long key = keys[p];
        const int m = (int)((key >> 12) & 15), n = (int)(key & 4095);
        const double mA = (double)acc[AS * p], mAA = (double)acc[AS * p + 1];
        out[(int64_t)s_cells[m] * stride + n] = pearson_from_moments<T>(mA, mAA, (double)acc[AS * p + 2], s_sb[m], s_sbb[m], (double)G);
        if (DUAL) out2[(int64_t)s_cells[m] * stride + n] = pearson_from_moments<T>(mA, mAA, (double)acc[AS * p + 3], s_sb2[m], s_sbb2[m], (double)G);
    }
}

// ---------------------------------------------------------------------------------------------
// Full kernel: every (c, i) pair.  C^2 * G transform evaluations -> VALU/transcendental-bound, so
// the job is to load each element once per tile and keep the inner loop register-resident:
// a 256-thread block owns a TC x TI = 16 x 64 tile of pairs and walks genes GK = 32 at a time;
// e[i] tile [64][33] (padded: conflict-free column reads), e[c]/d[c] tiles [16][32] broadcast.
// Thread (i = tid & 63, cg = tid >> 6) accumulates the three raw moments for 4 cells c.
constexpr int FULL_TC = 16, FULL_TI = 64, FULL_GK = 32;

template <typename T, int TR>
__global__ __launch_bounds__(256) void k_cdc_full(const T *__restrict__ e, const T *__restrict__ d, T *__restrict__ rm,
                                                   int C, int G, int64_t ld, int64_t cell0, int C_out, int64_t ld_rm,
                                                   T psc, int accumulate)
{
    __shared__ T ei[FULL_TI][FULL_GK + 1];
    __shared__ T ecs[FULL_TC][FULL_GK];
    __shared__ T dcs[FULL_TC][FULL_GK];
    __shared__ double sbs[FULL_TC][2];
    const int tid = threadIdx.x, il = tid & 63, cg = tid >> 6;
    const int i0 = blockIdx.x * FULL_TI, c0 = blockIdx.y * FULL_TC;
    T sA[4], sAA[4], sAb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { sA[k] = T(0); sAA[k] = T(0); sAb[k] = T(0); }
    // staging roles: e[i] tile = 64 rows x 32 genes -> thread loads rows (tid>>5) + 8*r, gene tid&31
    // (a wave reads two 128-B row segments per instruction); d/e[c] tile = 16 x 32 -> 2 elements each.
    const int sg = tid & 31, sr = tid >> 5;
    double sb = 0.0, sbb = 0.0;  // partial sums of d for row (sr) and (sr + 8), kept by the loader
    double sb2 = 0.0, sbb2 = 0.0;
    const T K = xform_shift<T, TR, VCY_RULES_FULL>(psc);
    for (int g0 = 0; g0 < G; g0 += FULL_GK) {
        const int glen = min(FULL_GK, G - g0);
        __syncthreads();
        const bool gok = sg < glen;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int row = sr + 8 * r, gi = i0 + row;
            ei[row][sg] = (gok && gi < C) ? e[(int64_t)gi * ld + g0 + sg] : T(0);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int row = sr + 8 * r;
            const int64_t gc = cell0 + c0 + row;
            const bool ok = gok && (c0 + row) < C_out;
            const T ev = ok ? e[gc * ld + g0 + sg] : T(0);
            const T dv = ok ? d[gc * ld + g0 + sg] : T(0);
            ecs[row][sg] = ev;
            dcs[row][sg] = dv;
            if (r == 0) { sb += (double)dv; sbb += (double)dv * (double)dv; }
            else { sb2 += (double)dv; sbb2 += (double)dv * (double)dv; }
        }
        __syncthreads();
        for (int g = 0; g < glen; ++g) {
            const T x = ei[il][g];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                T a = xform_s<T, TR, VCY_RULES_FULL>(x - ecs[cg * 4 + k][g], psc, K);
                sA[k] += a;
                sAA[k] = fma(a, a, sAA[k]);
                sAb[k] = fma(a, dcs[cg * 4 + k][g], sAb[k]);
            }
        }
    }
    // reduce the d-moments over the 32 loader lanes that share a row (half-wave groups)
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) {
        sb += __shfl_xor(sb, off, 64); sbb += __shfl_xor(sbb, off, 64);
        sb2 += __shfl_xor(sb2, off, 64); sbb2 += __shfl_xor(sbb2, off, 64);
    }
    if (sg == 0) { sbs[sr][0] = sb; sbs[sr][1] = sbb; sbs[sr + 8][0] = sb2; sbs[sr + 8][1] = sbb2; }
    __syncthreads();
    const int gi = i0 + il;
    if (gi < C) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int crow = c0 + cg * 4 + k;
            if (crow < C_out) {
                const T r = pearson_from_moments<T>((double)sA[k], (double)sAA[k], (double)sAb[k],
                                                    sbs[cg * 4 + k][0], sbs[cg * 4 + k][1], (double)G);
                T *p = rm + (int64_t)crow * ld_rm + gi;
                *p = accumulate ? (*p + r) : r;
            }
        }
    }
}

template <typename T>
__global__ void k_scatter_rows(const T *__restrict__ vals, const int32_t *__restrict__ ixs, T *__restrict__ rm,
                               int64_t total, int nrndm, int64_t ld_rm)
{
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = t / nrndm;
        atomicAdd(rm + c * ld_rm + ixs[t], vals[t]);
    }
}

// ---------------------------------------------------------------------------------------------
template <typename T, int TR, int RULES, int GC, int NV, bool DUAL>
static int launch_grouped(const void *e, const void *d, const void *d2, const int32_t *ixs, void *out, void *out2, const int32_t *order, int64_t G,
                          int64_t ld, int64_t cell0, int64_t C_out, int64_t d_row0, int64_t nrndm, double psc, hipStream_t st,
                          FuseArgs<T> fuse, const DevInfo &dev, bool *done)
{
    // Cells adjacent in the schedule order share neighbour rows out of LDS.  Neighbour lists wider than one workgroup's LDS
    // budget (8 x 256 pairs) are walked in column TILES of equal width, block -> (group, tile), each writing its own columns
    // of `out` (the reference default n_neighbors = C/5, sampled_fraction 0.3 gives nrndm = 3000: 12 tiles; rows of ixs
    // sorted by neighbour index make the tiles of adjacent cells overlap)
    constexpr int64_t TILE_MAX = 256;
    static_assert(GC * TILE_MAX <= 2048, "a row descriptor holds the first pair of the row in 11 bits");
    *done = false;
    const size_t budget_g = (size_t)(dev.lds_optin > 163840 ? 163840 : dev.lds_optin);
    int64_t ntiles = (nrndm + TILE_MAX - 1) / TILE_MAX, tile = 0;
    int npad = 2;
    size_t lds_g = 0;
    for (;; ++ntiles) {                                  // fewest equal-width tiles whose sort keys + accumulators fit (f64 needs narrower ones)
        tile = (nrndm + ntiles - 1) / ntiles;
        const int64_t maxpairs = GC * tile;
        for (npad = 2; npad < maxpairs; npad <<= 1) {}
        lds_g = grouped_lds_bytes<T>(GC, NV, DUAL, maxpairs, npad);
        if (lds_g <= budget_g || tile <= 16) break;
    }
    if (!(nrndm >= 8 && C_out >= 4 * GC && nrndm <= 0x7fffffff / 2 && lds_g <= budget_g)) return VCY_OK;
    auto kern = k_cdc_partial_grouped<T, TR, RULES, GC, NV, DUAL>;
    int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds_g);
    if (rc) return rc;
    // One workgroup per CU is resident (LDS), every group costs the same, so the last round of a launch would leave
    // most CUs idle for a whole group time (6250 groups on 256 CUs: 24.4 rounds; a 6250-cell shard of an 8-GPU run:
    // 3.05 rounds -> 4).  The groups beyond the last full round therefore run as the last blocks of the launch with their
    // neighbour lists cut into narrower tiles, as many (group, tile) blocks as there are CUs.
    const int64_t groups = (C_out + GC - 1) / GC, W = dev.cus > 0 ? dev.cus : 256;
    const int64_t full = groups >= W ? groups / W * W : 0;       // (from one round on: a 270-group launch is one round + a tiled tail, not two rounds)
    const int64_t c_main = full * GC, c_tail = C_out - c_main;
    int64_t tw = tile;
    if (c_tail > 0) {
        const int64_t left = groups - full;
        int64_t split = W / (left * ntiles);                       // how many pieces each base tile can be cut into
        if (split < 1) split = 1;
        tw = (tile + split - 1) / split;
        if (tw < 16) tw = tile < 16 ? tile : 16;
    }
    auto nblocks = [&](int64_t ncell, int64_t w) { return ncell > 0 ? ((ncell + GC - 1) / GC + 7) / 8 * 8 * ((nrndm + w - 1) / w) : (int64_t)0; };
    hipLaunchKernelGGL(kern, dim3((unsigned)(nblocks(c_main, tile) + nblocks(c_tail, tw))), dim3(1024), lds_g, st, (const T *)e, (const T *)d,
                       (const T *)d2, ixs, (T *)out, (T *)out2, order, (int)G, ld, cell0, d_row0, (int)c_main, (int)tile, (int)c_tail, (int)tw,
                       (int)nrndm, (int)nrndm, npad, (T)psc, fuse);
    VCY_LAUNCH_CHECK();
    *done = true;
    return VCY_OK;
}

// d2 / out2 non-null: the dual-control form (both correlations of a pair from one evaluation of A).
template <typename T, int TR, int RULES>
static int launch_partial(const void *e, const void *d, const void *d2, const int32_t *ixs, void *out, void *out2, const int32_t *order, int64_t G,
                          int64_t ld, int64_t cell0, int64_t C_out, int64_t d_row0, int64_t nrndm, double psc, hipStream_t st,
                          FuseArgs<T> fuse = FuseArgs<T>{nullptr, nullptr, nullptr, T(1), T(1)})
{
    constexpr int N = Vec<T>::N;
    DevInfo dev;
    int rc = device_info(&dev);
    if (rc) return rc;
    if (env_int("VCY_CDC_GROUP", GRP_GC) == GRP_GC) {            // VCY_CDC_GROUP=0: one cell per workgroup (A/B testing)
        bool done = false;
        if (d2) {
            if constexpr (sizeof(T) == 8) {
                // f64 dual control: 4 cells x 1024-gene chunks (three staged arrays of 8-byte elements: 96 KiB) - the single kernel's chunk
                // length, hence its order of summation (real correlations bit-identical to the single launch) and its ratio of
                // reduction work per element; 6 cells x 768 (VCY_CDC_DUAL_F64=0) shares rows better but pays 12-element chunks
                if (env_int("VCY_CDC_DUAL_F64", 1) == 1)
                    rc = launch_grouped<T, TR, RULES, GRP_GC_DUAL_F64, GRP_NV_F64, true>(e, d, d2, ixs, out, out2, order, G, ld, cell0, C_out, d_row0, nrndm, psc, st, fuse, dev, &done);
                else
                    rc = launch_grouped<T, TR, RULES, GRP_GC_DUAL, GRP_NV_DUAL, true>(e, d, d2, ixs, out, out2, order, G, ld, cell0, C_out, d_row0, nrndm, psc, st, fuse, dev, &done);
            } else
                rc = launch_grouped<T, TR, RULES, GRP_GC_DUAL, GRP_NV_DUAL, true>(e, d, d2, ixs, out, out2, order, G, ld, cell0, C_out, d_row0, nrndm, psc, st, fuse, dev, &done);
        } else if constexpr (sizeof(T) == 8)      // f64: 6 cells x 1024 genes (8 vectors per lane) measured 4.5 % faster than 8 x 768
            rc = launch_grouped<T, TR, RULES, GRP_GC_F64, GRP_NV_F64, false>(e, d, nullptr, ixs, out, nullptr, order, G, ld, cell0, C_out, d_row0, nrndm, psc, st, fuse, dev, &done);
        else
            rc = launch_grouped<T, TR, RULES, GRP_GC, GRP_NV, false>(e, d, nullptr, ixs, out, nullptr, order, G, ld, cell0, C_out, d_row0, nrndm, psc, st, fuse, dev, &done);
        if (rc || done) return rc;
    }
    if (d2) {   // small problems: the one-cell-per-workgroup kernel once per control (never fused: d2 is a materialised matrix)
        rc = launch_partial<T, TR, RULES>(e, d, nullptr, ixs, out, nullptr, order, G, ld, cell0, C_out, d_row0, nrndm, psc, st, fuse);
        if (rc) return rc;
        return launch_partial<T, TR, RULES>(e, d2, nullptr, ixs, out2, nullptr, order, G, ld, cell0, C_out, d_row0, nrndm, psc, st);
    }
    const int quantum = 64 * N;  // one wave-instruction worth of elements
    const size_t fixed = sizeof(T) * 3 * ((nrndm + 1) & ~1) + 32 * sizeof(double);
    // budget: stay under 150 KiB so one workgroup (16 waves) owns a CU; fewest chunks that fit
    const size_t budget = (size_t)(dev.lds_optin > 153600 ? 153600 : dev.lds_optin);
    if (fixed + 2 * quantum * sizeof(T) > budget)
        return fail(VCY_ERR_UNSUPPORTED, "%s: nrndm=%lld needs more LDS than the %lld-byte budget", "coldeltacor_partial", (long long)nrndm, (long long)budget);
    const int64_t max_chunk = (int64_t)((budget - fixed) / (2 * sizeof(T))) / quantum * quantum;
    const int64_t Gq = (G + quantum - 1) / quantum * quantum;
    const int64_t nchunks = (Gq + max_chunk - 1) / max_chunk;
    int64_t gchunk = ((Gq / quantum + nchunks - 1) / nchunks) * quantum;
    const size_t lds = fixed + 2 * gchunk * sizeof(T);
    auto kern = k_cdc_partial<T, TR, RULES>;
    rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
    if (rc) return rc;
    const int threads = (nrndm >= 16) ? 1024 : (nrndm >= 8 ? 512 : 256);
    hipLaunchKernelGGL(kern, dim3((unsigned)((C_out + 7) / 8 * 8)), dim3(threads), lds, st, (const T *)e, (const T *)d, ixs, (T *)out, order,
                       (int)G, ld, cell0, d_row0, (int)C_out, (int)nrndm, (int)gchunk, (T)psc, fuse);
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}

template <typename T>
static int dispatch_partial(const void *e, const void *d, const void *d2, const int32_t *ixs, void *out, void *out2, const int32_t *order, int64_t G,
                            int64_t ld, int64_t cell0, int64_t C_out, int64_t d_row0, int64_t nrndm, int transform, int rules,
                            double psc, hipStream_t st, FuseArgs<T> fuse = FuseArgs<T>{nullptr, nullptr, nullptr, T(1), T(1)})
{
#define VCY_CASE(TR, RU) \
    if (transform == TR && rules == RU) return launch_partial<T, TR, RU>(e, d, d2, ixs, out, out2, order, G, ld, cell0, C_out, d_row0, nrndm, psc, st, fuse);
    VCY_CASE(VCY_LINEAR, VCY_RULES_PARTIAL)
    VCY_CASE(VCY_LINEAR, VCY_RULES_FULL)
    VCY_CASE(VCY_SQRT, VCY_RULES_PARTIAL)
    VCY_CASE(VCY_SQRT, VCY_RULES_FULL)
    VCY_CASE(VCY_LOG10, VCY_RULES_PARTIAL)
    VCY_CASE(VCY_LOG10, VCY_RULES_FULL)
    if constexpr (std::is_same<T, float>::value) { VCY_CASE(VCY_SQRT, VCY_RULES_PARTIAL_NOPSC) }
#undef VCY_CASE
    if (rules == VCY_RULES_PARTIAL_NOPSC) return fail(VCY_ERR_INVALID, "%s: VCY_RULES_PARTIAL_NOPSC is defined for VCY_SQRT on VCY_F32 only", "coldeltacor_partial");
    return fail(VCY_ERR_INVALID, "%s: bad transform/rules", "coldeltacor_partial");
}

template <typename T>
static int dispatch_full(const void *e, const void *d, void *rm, int64_t C, int64_t G, int64_t ld, int64_t cell0,
                         int64_t C_out, int64_t ld_rm, int transform, double psc, int accumulate, hipStream_t st)
{
    dim3 grid((unsigned)((C + FULL_TI - 1) / FULL_TI), (unsigned)((C_out + FULL_TC - 1) / FULL_TC));
#define VCY_CASE(TR)                                                                                                        \
    if (transform == TR) {                                                                                                  \
        hipLaunchKernelGGL((k_cdc_full<T, TR>), grid, dim3(256), 0, st, (const T *)e, (const T *)d, (T *)rm, (int)C, (int)G, \
                           ld, cell0, (int)C_out, ld_rm, (T)psc, accumulate);                                              \
        VCY_LAUNCH_CHECK();                                                                                                 \
        return VCY_OK;                                                                                                      \
    }
    VCY_CASE(VCY_LINEAR)
    VCY_CASE(VCY_SQRT)
    VCY_CASE(VCY_LOG10)
#undef VCY_CASE
    return fail(VCY_ERR_INVALID, "%s: bad transform", "coldeltacor_full");
}

}  // namespace vcy

using namespace vcy;

static int check_partial_args(const char *who, const void *e, const void *d, const int32_t *ixs, const void *out, int64_t C, int64_t G, int64_t ld,
                              int64_t cell0, int64_t C_out, int64_t d_row0, int64_t nrndm, int dtype)
{
    if (!(e && d && ixs && out)) return fail(VCY_ERR_INVALID, "%s: null pointer", who);
    if (!(C > 0 && G > 0 && nrndm > 0 && C_out > 0 && cell0 >= 0 && cell0 + C_out <= C)) return fail(VCY_ERR_INVALID, "%s: bad shape", who);
    if (ld < G) return fail(VCY_ERR_INVALID, "%s: ld < G", who);
    if (!(d_row0 >= 0 && d_row0 <= cell0)) return fail(VCY_ERR_INVALID, "%s: d must cover cells cell0..cell0+C_out-1", who);
    if (!(dtype == VCY_F32 || dtype == VCY_F64)) return fail(VCY_ERR_INVALID, "%s: bad dtype", who);
    if (ld % (dtype == VCY_F32 ? 4 : 2) != 0) return fail(VCY_ERR_INVALID, "%s: ld must keep rows 16-byte aligned", who);
    if (((uintptr_t)e % 16) || ((uintptr_t)d % 16)) return fail(VCY_ERR_INVALID, "%s: e/d must be 16-byte aligned", who);
    return VCY_OK;
}

extern "C" int vcy_coldeltacor_partial(const void *e, const void *d, const int32_t *ixs, void *out, const int32_t *order,
                                       int64_t C, int64_t G, int64_t ld, int64_t cell0, int64_t C_out, int64_t d_row0,
                                       int64_t nrndm, int transform, int rules, double psc, int dtype, vcy_stream stream)
{
    int rc = check_partial_args("coldeltacor_partial", e, d, ixs, out, C, G, ld, cell0, C_out, d_row0, nrndm, dtype);
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    if (dtype == VCY_F32) return dispatch_partial<float>(e, d, nullptr, ixs, out, nullptr, order, G, ld, cell0, C_out, d_row0, nrndm, transform, rules, psc, st);
    return dispatch_partial<double>(e, d, nullptr, ixs, out, nullptr, order, G, ld, cell0, C_out, d_row0, nrndm, transform, rules, psc, st);
}

extern "C" int vcy_coldeltacor_partial_dual(const void *e, const void *d, const void *d_rndm, const int32_t *ixs, void *out, void *out_rndm,
                                            const int32_t *order, int64_t C, int64_t G, int64_t ld, int64_t cell0, int64_t C_out, int64_t d_row0,
                                            int64_t nrndm, int transform, int rules, double psc, int dtype, vcy_stream stream)
{
    int rc = check_partial_args("coldeltacor_partial_dual", e, d, ixs, out, C, G, ld, cell0, C_out, d_row0, nrndm, dtype);
    if (rc) return rc;
    VCY_REQUIRE(d_rndm && out_rndm && out_rndm != out, "coldeltacor_partial_dual: d_rndm / out_rndm missing");
    VCY_REQUIRE((uintptr_t)d_rndm % 16 == 0, "coldeltacor_partial_dual: d_rndm must be 16-byte aligned");
    hipStream_t st = as_stream(stream);
    if (dtype == VCY_F32) return dispatch_partial<float>(e, d, d_rndm, ixs, out, out_rndm, order, G, ld, cell0, C_out, d_row0, nrndm, transform, rules, psc, st);
    return dispatch_partial<double>(e, d, d_rndm, ixs, out, out_rndm, order, G, ld, cell0, C_out, d_row0, nrndm, transform, rules, psc, st);
}

extern "C" int vcy_coldeltacor_partial_fused(const void *Sx_sz, const void *Ux_sz, const float *gamma, const float *q, const int32_t *ixs,
                                            void *out, const int32_t *order, int64_t C, int64_t G, int64_t ld, int64_t cell0, int64_t C_out,
                                            int64_t u_row0, int64_t nrndm, int transform, int rules, double psc, double dt_shift,
                                            double used_dt, int dtype, vcy_stream stream)
{
    int rc = check_partial_args("coldeltacor_partial_fused", Sx_sz, Ux_sz, ixs, out, C, G, ld, cell0, C_out, u_row0, nrndm, dtype);
    if (rc) return rc;
    VCY_REQUIRE(gamma, "coldeltacor_partial_fused: null pointer");
    hipStream_t st = as_stream(stream);
    if (dtype == VCY_F32) {
        FuseArgs<float> f{(const float *)Ux_sz, gamma, q, (float)dt_shift, (float)used_dt};
        return dispatch_partial<float>(Sx_sz, Ux_sz, nullptr, ixs, out, nullptr, order, G, ld, cell0, C_out, u_row0, nrndm, transform, rules, psc, st, f);
    }
    FuseArgs<double> f{(const double *)Ux_sz, gamma, q, dt_shift, used_dt};
    return dispatch_partial<double>(Sx_sz, Ux_sz, nullptr, ixs, out, nullptr, order, G, ld, cell0, C_out, u_row0, nrndm, transform, rules, psc, st, f);
}

// Stage C folded in AND the randomised control in the same pass: d[c] is evaluated from Ux, gamma, q while it is staged
// (as in vcy_coldeltacor_partial_fused), d_rndm is the materialised transform of the permuted delta_S.
extern "C" int vcy_coldeltacor_partial_fused_dual(const void *Sx_sz, const void *Ux_sz, const float *gamma, const float *q, const void *d_rndm,
                                                 const int32_t *ixs, void *out, void *out_rndm, const int32_t *order, int64_t C, int64_t G, int64_t ld,
                                                 int64_t cell0, int64_t C_out, int64_t u_row0, int64_t nrndm, int transform, int rules, double psc,
                                                 double dt_shift, double used_dt, int dtype, vcy_stream stream)
{
    int rc = check_partial_args("coldeltacor_partial_fused_dual", Sx_sz, Ux_sz, ixs, out, C, G, ld, cell0, C_out, u_row0, nrndm, dtype);
    if (rc) return rc;
    VCY_REQUIRE(gamma && d_rndm && out_rndm && out_rndm != out, "coldeltacor_partial_fused_dual: null pointer");
    VCY_REQUIRE((uintptr_t)d_rndm % 16 == 0, "coldeltacor_partial_fused_dual: d_rndm must be 16-byte aligned");
    hipStream_t st = as_stream(stream);
    if (dtype == VCY_F32) {
        FuseArgs<float> f{(const float *)Ux_sz, gamma, q, (float)dt_shift, (float)used_dt};
        return dispatch_partial<float>(Sx_sz, Ux_sz, d_rndm, ixs, out, out_rndm, order, G, ld, cell0, C_out, u_row0, nrndm, transform, rules, psc, st, f);
    }
    FuseArgs<double> f{(const double *)Ux_sz, gamma, q, dt_shift, used_dt};
    return dispatch_partial<double>(Sx_sz, Ux_sz, d_rndm, ixs, out, out_rndm, order, G, ld, cell0, C_out, u_row0, nrndm, transform, rules, psc, st, f);
}

extern "C" int vcy_coldeltacor_full(const void *e, const void *d, void *rm, int64_t C, int64_t G, int64_t ld, int64_t cell0,
                                    int64_t C_out, int64_t ld_rm, int transform, double psc, int accumulate, int dtype,
                                    vcy_stream stream)
{
    VCY_REQUIRE(e && d && rm, "coldeltacor_full: null pointer");
    VCY_REQUIRE(C > 0 && G > 0 && C_out > 0 && cell0 >= 0 && cell0 + C_out <= C && ld >= G && ld_rm >= C, "coldeltacor_full: bad shape");
    VCY_REQUIRE(dtype == VCY_F32 || dtype == VCY_F64, "coldeltacor_full: bad dtype");
    hipStream_t st = as_stream(stream);
    if (dtype == VCY_F32) return dispatch_full<float>(e, d, rm, C, G, ld, cell0, C_out, ld_rm, transform, psc, accumulate, st);
    return dispatch_full<double>(e, d, rm, C, G, ld, cell0, C_out, ld_rm, transform, psc, accumulate, st);
}

extern "C" int vcy_scatter_rows(const void *vals, const int32_t *ixs, void *rm, int64_t C_out, int64_t nrndm, int64_t ld_rm,
                                int dtype, vcy_stream stream)
{
    VCY_REQUIRE(vals && ixs && rm && C_out > 0 && nrndm > 0, "scatter_rows: bad arguments");
    const int64_t total = C_out * nrndm;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipStream_t st = as_stream(stream);
    if (dtype == VCY_F32) hipLaunchKernelGGL(k_scatter_rows<float>, dim3(blocks), dim3(256), 0, st, (const float *)vals, ixs, (float *)rm, total, (int)nrndm, ld_rm);
    else if (dtype == VCY_F64) hipLaunchKernelGGL(k_scatter_rows<double>, dim3(blocks), dim3(256), 0, st, (const double *)vals, ixs, (double *)rm, total, (int)nrndm, ld_rm);
    else return fail(VCY_ERR_INVALID, "%s: bad dtype", "scatter_rows");
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}
