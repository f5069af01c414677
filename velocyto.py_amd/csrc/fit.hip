// fit.hip -- stage B: per-gene gamma fits and the per-gene order statistics behind their weights.
//
// Reference: estimation.fit_slope* / _fit1_slope* (estimation.py:173-366) loop over genes in
// Python and call scipy optimisers on each gene's (C,) vectors; VelocytoLoom.fit_gammas
// (analysis.py:1179-1219) builds the weights from np.percentile(axis=1).
// Here a gene is a COLUMN of the cells-major matrices.  One streaming pass accumulates all
// per-gene moments (fp64) for a block of cells per workgroup, lanes mapped to adjacent genes so
// every load is a coalesced row segment; a second tiny kernel reduces the per-block partials in
// a fixed order (deterministic, no atomics) and solves each gene's least-squares problem in
// closed form -- the problems the reference hands to nnls / Brent / L-BFGS-B are 1- or
// 2-parameter convex quadratics over a box.  HBM-bound: 2 * G * s bytes per cell (plain fit).
#include <math.h>
#include "common.h"

namespace vcy {

constexpr int FIT_CB = 32;        // cell blocks (partials per gene)
constexpr int FIT_NMOM = 10;      // Sx Sy Sxx Sxy Syy | Sw Swx Swy Swxx Swxy

// ---------------------------------------------------------------- plain fit_slope moments
// thread = one 16-byte gene vector; block = 256 threads; grid = (gene tiles, FIT_CB).
template <typename T>
__global__ __launch_bounds__(256) void k_moments_plain(const T *__restrict__ Y, const T *__restrict__ X, double *__restrict__ part,
                                                        int C, int G, int64_t ld)
{
    using V = typename Vec<T>::type;
    constexpr int N = Vec<T>::N;
    const int v = blockIdx.x * blockDim.x + threadIdx.x;   // vector index along genes
    const int g = v * N;
    if (g >= G) return;
    const int cb = blockIdx.y;
    const int per = (C + FIT_CB - 1) / FIT_CB;
    const int c0 = cb * per, c1 = min(C, c0 + per);
    double sxx[N], sxy[N], syy[N];
#pragma unroll
    for (int k = 0; k < N; ++k) { sxx[k] = 0; sxy[k] = 0; syy[k] = 0; }
    int c = c0;
    for (; c + 3 < c1; c += 4) {
        V xv[4], yv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            xv[u] = reinterpret_cast<const V *>(X + (int64_t)(c + u) * ld)[v];
            yv[u] = reinterpret_cast<const V *>(Y + (int64_t)(c + u) * ld)[v];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const T *xp = reinterpret_cast<const T *>(&xv[u]);
            const T *yp = reinterpret_cast<const T *>(&yv[u]);
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const double x = xp[k], y = yp[k];
                sxx[k] = fma(x, x, sxx[k]); sxy[k] = fma(x, y, sxy[k]); syy[k] = fma(y, y, syy[k]);
            }
        }
    }
    for (; c < c1; ++c) {
        const V xv = reinterpret_cast<const V *>(X + (int64_t)c * ld)[v];
        const V yv = reinterpret_cast<const V *>(Y + (int64_t)c * ld)[v];
        const T *xp = reinterpret_cast<const T *>(&xv);
        const T *yp = reinterpret_cast<const T *>(&yv);
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const double x = xp[k], y = yp[k];
            sxx[k] = fma(x, x, sxx[k]); sxy[k] = fma(x, y, sxy[k]); syy[k] = fma(y, y, syy[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (g + k < G) {
            double *p = part + ((int64_t)cb * 3) * G + g + k;
            p[0] = sxx[k]; p[(int64_t)G] = sxy[k]; p[2 * (int64_t)G] = syy[k];
        }
    }
}

// fixed-order reduction of the FIT_CB per-block partials -> moments (3, G)
__global__ void k_fit_slope_reduce(const double *__restrict__ part, double *__restrict__ mom, int G)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    double sxx = 0, sxy = 0, syy = 0;
    for (int cb = 0; cb < FIT_CB; ++cb) {
        const double *p = part + ((int64_t)cb * 3) * G + g;
        sxx += p[0]; sxy += p[(int64_t)G]; syy += p[2 * (int64_t)G];
    }
    mom[g] = sxx; mom[(int64_t)G + g] = sxy; mom[2 * (int64_t)G + g] = syy;
}

// estimation.py:173-188: not any(x) -> NaN; not any(y) -> 0; else nnls == max(0, <x,y>/<x,x>).
__global__ void k_fit_slope_final(const double *__restrict__ mom, float *__restrict__ gamma, int G)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const double sxx = mom[g], sxy = mom[(int64_t)G + g], syy = mom[2 * (int64_t)G + g];
    double m;
    if (!(sxx > 0)) m = NAN;
    else if (!(syy > 0)) m = 0.0;
    else { m = sxy / sxx; if (m < 0) m = 0; }
    gamma[g] = (float)m;
}

// ---------------------------------------------------------------- weighted moments
// Z value used by the "maxmin*" weights (analysis.py:1203-1206): evaluated in T so that the
// thresholds (k_gene_quantiles) and the weights (here) see bit-identical numbers.
template <typename T> __device__ __forceinline__ T zvalue(T m, T m2, T a, T b, bool two)
{
    return two ? (m / a + m2 / b) : m;   // true divisions, like Sx/denom_Sx + Ux/denom_Ux
}

template <typename T, int WMODE>
__global__ __launch_bounds__(256) void k_moments_weighted(const T *__restrict__ Y, const T *__restrict__ X, const T *__restrict__ W,
                                                           const T *__restrict__ M, const T *__restrict__ M2,
                                                           const double *__restrict__ scale_a, const double *__restrict__ scale_b,
                                                           const double *__restrict__ down, const double *__restrict__ up,
                                                           double *__restrict__ part, int C, int G, int64_t ld)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;   // lane <-> gene: coalesced 4/8-byte row segments
    if (g >= G) return;
    const int cb = blockIdx.y;
    const int per = (C + FIT_CB - 1) / FIT_CB;
    const int c0 = cb * per, c1 = min(C, c0 + per);
    const bool two = (WMODE == 1) && scale_a != nullptr;
    T den_a = T(1), den_b = T(1);
    double dn = 0, upv = 0;        // thresholds stay fp64: compared against (double)z like the reference's X <= down
    if (WMODE == 1) {
        if (two) { den_a = (T)scale_a[g]; den_b = (T)scale_b[g]; }
        dn = down[g]; upv = up[g];
    }
    double sx = 0, sy = 0, sxx = 0, sxy = 0, syy = 0, sw = 0, swx = 0, swy = 0, swxx = 0, swxy = 0;
#pragma unroll 4
    for (int c = c0; c < c1; ++c) {
        const int64_t o = (int64_t)c * ld + g;
        const double x = X[o], y = Y[o];
        double w;
        if (WMODE == 0) w = W[o];
        else if (WMODE == 1) {
            const double z = (double)zvalue<T>(M[o], two ? M2[o] : T(0), den_a, den_b, two);
            w = (z <= dn || z >= upv) ? 1.0 : 0.0;
        } else w = 1.0;
        sx += x; sy += y; sxx = fma(x, x, sxx); sxy = fma(x, y, sxy); syy = fma(y, y, syy);
        const double wx = w * x;
        sw += w; swx += wx; swy = fma(w, y, swy); swxx = fma(wx, x, swxx); swxy = fma(wx, y, swxy);
    }
    double *p = part + ((int64_t)cb * FIT_NMOM) * G + g;
    const int64_t s = G;
    p[0] = sx; p[s] = sy; p[2 * s] = sxx; p[3 * s] = sxy; p[4 * s] = syy;
    p[5 * s] = sw; p[6 * s] = swx; p[7 * s] = swy; p[8 * s] = swxx; p[9 * s] = swxy;
}

// Exact minimiser of  sum w (x m + q - y)^2  over [lo_m,hi_m] x [lo_q,hi_q] from the weighted
// moments: interior stationary point if feasible, else the best clipped edge minimum (convex).
__device__ inline void box_wls2(double sw, double sx, double sy, double sxx, double sxy, double lo_m, double hi_m,
                                double lo_q, double hi_q, double &m_out, double &q_out)
{
    auto f = [&](double m, double q) { return m * m * sxx + q * q * sw + 2 * m * q * sx - 2 * m * sxy - 2 * q * sy; };
    const double det = sxx * sw - sx * sx;
    if (det > 0) {
        const double m = (sxy * sw - sx * sy) / det, q = (sxx * sy - sx * sxy) / det;
        if (m >= lo_m && m <= hi_m && q >= lo_q && q <= hi_q) { m_out = m; q_out = q; return; }
    }
    double best = INFINITY;
    const double qs[2] = {lo_q, hi_q}, ms[2] = {lo_m, hi_m};
    for (int t = 0; t < 2; ++t) {
        const double q = qs[t];
        double m = sxx > 0 ? (sxy - q * sx) / sxx : lo_m;
        m = fmin(fmax(m, lo_m), hi_m);
        const double v = f(m, q);
        if (v < best) { best = v; m_out = m; q_out = q; }
    }
    for (int t = 0; t < 2; ++t) {
        const double m = ms[t];
        double q = sw > 0 ? (sy - m * sx) / sw : lo_q;
        q = fmin(fmax(q, lo_q), hi_q);
        const double v = f(m, q);
        if (v < best) { best = v; m_out = m; q_out = q; }
    }
}

// fit_offset=1, box_q=1 : _fit1_slope_weighted_offset (estimation.py:212-241), bounds
//                          gamma in [lo_gamma, up_gamma], q in [0, 2*sum(yw)/sum(w)]
// fit_offset=1, box_q=0 : _fit1_slope_offset (estimation.py:244-264) unconstrained OLS with intercept
// fit_offset=0          : _fit1_slope_weighted (estimation.py:191-209) / fixperc_q branches: gamma in
//                          [lo_gamma, up_gamma] with the offset fixed to q_fixed[g]
// R2 (estimation.py:323-331, 355-363) is unweighted, -1e16 when non-finite.
__global__ void k_fit_weighted_final(const double *__restrict__ part, int fit_offset, int box_q, double lo_gamma,
                                     double up_gamma_default, const double *__restrict__ up_gamma, const double *__restrict__ q_fixed,
                                     float *__restrict__ gamma, float *__restrict__ qout, float *__restrict__ R2, int C, int G)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    double mo[FIT_NMOM];
    for (int k = 0; k < FIT_NMOM; ++k) mo[k] = 0;
    for (int cb = 0; cb < FIT_CB; ++cb)
        for (int k = 0; k < FIT_NMOM; ++k) mo[k] += part[((int64_t)cb * FIT_NMOM + k) * G + g];
    const double sx = mo[0], sy = mo[1], sxx = mo[2], sxy = mo[3], syy = mo[4];
    const double sw = mo[5], swx = mo[6], swy = mo[7], swxx = mo[8], swxy = mo[9];
    const double n = (double)C;
    const double hi = up_gamma ? up_gamma[g] : up_gamma_default;
    double m, q;
    if (!(sxx > 0)) { m = NAN; q = 0.0; }
    else if (!(syy > 0)) { m = 0.0; q = 0.0; }
    else if (fit_offset) {
        if (box_q) box_wls2(sw, swx, swy, swxx, swxy, lo_gamma, hi, 0.0, 2.0 * swy / sw, m, q);
        else {
            const double det = swxx * sw - swx * swx;
            m = (swxy * sw - swx * swy) / det;
            q = (swy - m * swx) / sw;
        }
    } else {
        q = q_fixed ? q_fixed[g] : 0.0;
        m = swxx > 0 ? (swxy - q * swx) / swxx : lo_gamma;
        m = fmin(fmax(m, lo_gamma), hi);
    }
    gamma[g] = (float)m;
    if (qout) qout[g] = (float)q;
    if (R2) {
        // the reference evaluates R2 with the optimiser's fp64 (m, q), before the float32 store (estimation.py:351-359)
        const double ssres = m * m * sxx + n * q * q + syy + 2 * m * q * sx - 2 * m * sxy - 2 * q * sy;
        const double sstot = syy - sy * sy / n;
        const double r2 = 1.0 - ssres / sstot;
        R2[g] = isfinite(r2) ? (float)r2 : -1e16f;
    }
}

// ---------------------------------------------------------------- per-gene order statistics
// Step 1: gene-major key matrix Z (G, C): 64x64 LDS tile transpose of zvalue(M, M2).
// Optional mask (conditional percentiles of estimation.py:200-202, 222, 229-231, 255):
//   mask_mode 1 keeps cells with mask_src[c,g] >  mask_thr[g]   (y[x > percentile(x, 90)])
//   mask_mode 2 keeps cells with mask_src[c,g] <= mask_thr[g]   (y[x <= percentile(x, 1)])
// masked-out entries become +inf, sort last, and are not counted by k_gene_quantiles.
template <typename T>
__global__ __launch_bounds__(256) void k_build_z(const T *__restrict__ M, const T *__restrict__ M2, const double *__restrict__ scale_a,
                                                  const double *__restrict__ scale_b, const T *__restrict__ mask_src,
                                                  const double *__restrict__ mask_thr, int mask_mode, T *__restrict__ Z, int C, int G,
                                                  int64_t ld)
{
    __shared__ T tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c0 = blockIdx.y * 64, g0 = blockIdx.x * 64;
    const bool two = scale_a != nullptr;
    const int g = g0 + tx;
    T den_a = T(1), den_b = T(1);
    if (two && g < G) { den_a = (T)scale_a[g]; den_b = (T)scale_b[g]; }
    for (int j = ty; j < 64; j += 4) {
        const int c = c0 + j;
        T z = T(0);
        if (c < C && g < G) {
            const int64_t o = (int64_t)c * ld + g;
            z = zvalue<T>(M[o], two ? M2[o] : T(0), den_a, den_b, two);
            if (mask_mode) {
                const double mv = (double)mask_src[o], th = mask_thr[g];
                const bool keep = mask_mode == 1 ? (mv > th) : (mv <= th);
                if (!keep) z = (T)INFINITY;
            }
        }
        tile[j][tx] = z;
    }
    __syncthreads();
    for (int j = ty; j < 64; j += 4) {
        const int gg = g0 + j, cc = c0 + tx;
        if (gg < G && cc < C) Z[(int64_t)gg * C + cc] = tile[tx][j];
    }
}

struct QArgs { double q[16]; };   // percentiles travel as kernel arguments (no host->device copy)

template <typename T> struct Key;
template <> struct Key<float> {
    using U = uint32_t;
    static __device__ __forceinline__ U enc(float f) { U u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
    static __device__ __forceinline__ float dec(U u) { u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u; return __uint_as_float(u); }
};
template <> struct Key<double> {
    using U = uint64_t;
    static __device__ __forceinline__ U enc(double f) { U u = (U)__double_as_longlong(f); return (u >> 63) ? ~u : (u | 0x8000000000000000ull); }
    static __device__ __forceinline__ double dec(U u) { u = (u >> 63) ? (u & 0x7fffffffffffffffull) : ~u; return __longlong_as_double((long long)u); }
};

// One histogram increment per matching lane.  Expression rows are mostly exact zeros and a row's values share a few exponents, so the
// lanes of a wave mostly ask for the SAME bin - and LDS atomics on one address serialise, 64 deep when a whole wave holds zeros (the 2nd
// percentile of a 60 %-zero matrix cost twice the 98th).  The lanes that share the first matching lane's digit are therefore folded
// into one atomic carrying their count (VCY_QPEEL rounds of it); only the lanes left over add one each.
#ifndef VCY_QPEEL
#define VCY_QPEEL 1
#endif
__device__ __forceinline__ void hist_add(unsigned *hist, unsigned bin, bool match)
{
    const int lane = (int)(threadIdx.x & 63);
#pragma unroll
    for (int round = 0; round < VCY_QPEEL; ++round) {
        const unsigned long long m = __ballot(match);
        if (m == 0) return;
        const int leader = __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1);
        const unsigned lead_bin = (unsigned)__builtin_amdgcn_readlane((int)bin, leader);
        const bool same = match && bin == lead_bin;
        const unsigned long long ms = __ballot(same);
        if (lane == leader) atomicAdd(&hist[lead_bin], (unsigned)__popcll(ms));
        match = match && !same;
    }
    if (match) atomicAdd(&hist[bin], 1u);
}

// Step 2: one workgroup per gene: MSB-first 8-bit radix select of rank `r` over the gene's C keys
// (sizeof(key) passes, 256-bin LDS histogram; the row is L2-resident after the first pass), then one
// more pass for the next order statistic (count <= v, min > v).  numpy.percentile's default
// 'linear' rule: h = (C-1) q/100, lo = floor(h), t = h - lo,
//   t < 0.5 ? v[lo] + (v[lo+1]-v[lo]) t : v[lo+1] - (v[lo+1]-v[lo]) (1-t)      (numpy _lerp)
template <typename T>
__global__ __launch_bounds__(256) void k_gene_quantiles(const T *__restrict__ Z, QArgs qa, int nq,
                                                         double *__restrict__ out, int C, int G, int masked)
{
    using U = typename Key<T>::U;
    constexpr int PASSES = sizeof(U);
    __shared__ unsigned hist[256];
    __shared__ U s_prefix;
    __shared__ unsigned s_rank;
    __shared__ unsigned s_cnt_le;
    __shared__ unsigned long long s_min_gt;
    const int g = blockIdx.x, tid = threadIdx.x;
    const T *row = Z + (int64_t)g * C;
    int nvalid = C;
    if (masked) {   // entries removed by the mask were stored as +inf
        if (tid == 0) s_cnt_le = 0;
        __syncthreads();
        unsigned cnt = 0;
        for (int c = tid; c < C; c += 256) cnt += (row[c] < (T)INFINITY) ? 1u : 0u;
        cnt = wave_sum(cnt);
        if ((tid & 63) == 0) atomicAdd(&s_cnt_le, cnt);
        __syncthreads();
        nvalid = (int)s_cnt_le;
        __syncthreads();
    }
    for (int qi = 0; qi < nq; ++qi) {
        if (nvalid == 0) { if (tid == 0) out[(int64_t)qi * G + g] = NAN; continue; }
        const double h = __dmul_rn((double)(nvalid - 1), qa.q[qi] / 100.0);   // numpy: (n-1) * (q/100)
        const int lo = (int)floor(h);
        const double t = h - lo;
        U prefix = 0;
        unsigned rank = (unsigned)lo;   // 0-based rank among elements matching the prefix so far
        for (int pass = 0; pass < PASSES; ++pass) {
            const int shift = 8 * (PASSES - 1 - pass);
            hist[tid] = 0;
            __syncthreads();
            for (int c = tid; c < C; c += 256) {
                const U k = Key<T>::enc(row[c]);
                const bool match = (pass == 0) || ((k >> ((shift + 8) & (8 * PASSES - 1))) == (prefix >> ((shift + 8) & (8 * PASSES - 1))));
                hist_add(hist, (unsigned)((k >> shift) & 0xff), match);
            }
            __syncthreads();
            if (tid < 64) {   // one wave scans the 256 bins: 4 bins per lane
                unsigned h0 = hist[tid * 4], h1 = hist[tid * 4 + 1], h2 = hist[tid * 4 + 2], h3 = hist[tid * 4 + 3];
                unsigned tot = h0 + h1 + h2 + h3, incl = tot;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    unsigned o = __shfl_up(incl, off, 64);
                    if (tid >= off) incl += o;
                }
                const unsigned excl = incl - tot;
                if (rank >= excl && rank < incl) {   // exactly one lane
                    unsigned r = rank - excl;
                    int d;
                    if (r < h0) d = 0;
                    else if ((r -= h0) < h1) d = 1;
                    else if ((r -= h1) < h2) d = 2;
                    else { r -= h2; d = 3; }
                    s_prefix = prefix | ((U)(tid * 4 + d) << shift);
                    s_rank = r;
                }
            }
            __syncthreads();
            prefix = s_prefix;
            rank = s_rank;
        }
        const T vlo = Key<T>::dec(prefix);
        // next order statistic
        if (tid == 0) { s_cnt_le = 0; s_min_gt = ~0ull; }
        __syncthreads();
        unsigned cnt = 0;
        unsigned long long mg = ~0ull;
        for (int c = tid; c < C; c += 256) {
            const U k = Key<T>::enc(row[c]);
            if (k <= prefix) ++cnt;
            else if ((unsigned long long)k < mg) mg = (unsigned long long)k;
        }
        cnt = wave_sum(cnt);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(mg, off, 64);
            mg = o < mg ? o : mg;
        }
        if ((tid & 63) == 0) { atomicAdd(&s_cnt_le, cnt); atomicMin(&s_min_gt, mg); }
        __syncthreads();
        if (tid == 0) {
#pragma clang fp contract(off)   // hipcc would fuse the mul/add below even through the *_rn intrinsics
            T vhi = vlo;
            if (lo + 1 < nvalid && s_cnt_le < (unsigned)(lo + 2)) vhi = Key<T>::dec((U)s_min_gt);
            // numpy _lerp, without fma contraction so the rounding matches numpy's mul-then-add
            const double a = (double)vlo, b = (double)vhi, diff = __dsub_rn(b, a);
            double r = __dadd_rn(a, __dmul_rn(diff, t));
            if (t >= 0.5) r = __dsub_rn(b, __dmul_rn(diff, __dsub_rn(1.0, t)));
            if (t == 0.0) r = a;
            out[(int64_t)qi * G + g] = r;
        }
        __syncthreads();
    }
}

// Step 2, register-resident (round 3): the same selection with the gene's C keys held in the REGISTERS of a 1024-thread workgroup
// (NPT keys per thread, C <= 1024 NPT).  k_gene_quantiles re-reads the 200 KB row of a gene from memory on every radix pass - ten
// passes for two percentiles, and 2048 co-resident rows are 400 MB, far beyond L2: 60 GB of re-reads for a 6 GB matrix, 17 ms at
// 50 000 x 30 000.  Here the row is read ONCE (coalesced: thread t takes elements t, t + 1024, ...), every pass walks registers;
// histograms, scans and the interpolation are the ones of k_gene_quantiles, so the results are the same to the bit.
template <typename T, int NPT>
__global__ __launch_bounds__(1024) void k_gene_quantiles_reg(const T *__restrict__ Z, QArgs qa, int nq,
                                                              double *__restrict__ out, int C, int G, int masked)
{
    using U = typename Key<T>::U;
    constexpr int PASSES = sizeof(U);
    constexpr U KMAX = ~(U)0;                                // padding key: sorts last, never reaches a rank < nvalid
    constexpr int COPIES = 32, CSTRIDE = 257;                // lane l counts in copy l % 32; copy c starts at bank c, so equal digits of
    __shared__ unsigned hcopy[COPIES * CSTRIDE];             // different lanes fall in different banks (two lanes per word at worst)
    __shared__ unsigned hist[256];                           // the copies folded
    __shared__ unsigned hist0[256];                          // ... of the first digit: the same for every percentile
    __shared__ U s_prefix;
    __shared__ unsigned s_rank;
    __shared__ unsigned s_cnt_le;
    __shared__ unsigned long long s_min_gt;
    const int g = blockIdx.x, tid = threadIdx.x;
    const T *row = Z + (int64_t)g * C;
    U key[NPT];
    unsigned nfin = 0;
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        // branch-free (a load inside a branch makes hipcc wait for it inside the branch: NPT serialised round trips)
        const int c = tid + 1024 * i;
        const T v = row[c < C ? c : C - 1];
        key[i] = c < C ? Key<T>::enc(v) : KMAX;
        nfin += (c < C && v < (T)INFINITY) ? 1u : 0u;
    }
    for (int i = tid; i < COPIES * CSTRIDE; i += 1024) hcopy[i] = 0;
    if (tid < 256) hist[tid] = 0;
    if (tid == 0) s_cnt_le = 0;
    __syncthreads();
    int nvalid = C;
    if (masked) {   // entries removed by the mask were stored as +inf
        const unsigned cnt = wave_sum(nfin);
        if ((tid & 63) == 0) atomicAdd(&s_cnt_le, cnt);
        __syncthreads();
        nvalid = (int)s_cnt_le;
        __syncthreads();
    }
    unsigned *mine = hcopy + (tid & (COPIES - 1)) * CSTRIDE;
    for (int qi = 0; qi < nq; ++qi) {
        if (nvalid == 0) { if (tid == 0) out[(int64_t)qi * G + g] = NAN; continue; }
        const double h = __dmul_rn((double)(nvalid - 1), qa.q[qi] / 100.0);   // numpy: (n-1) * (q/100)
        const int lo = (int)floor(h);
        const double t = h - lo;
        U prefix = 0;
        unsigned rank = (unsigned)lo;
        // (the pass loop stays rolled: unrolled, hipcc hoists every key's digit of every pass out of the percentile loop -
        // PASSES x NPT more live registers, kilobytes of scratch)
#pragma unroll 1
        for (int pass = 0; pass < PASSES; ++pass) {
            const int shift = 8 * (PASSES - 1 - pass);
            const bool counted = pass == 0 && qi > 0;      // the first digit's histogram is kept from the first percentile
            if (!counted) {
#pragma unroll
                for (int i = 0; i < NPT; ++i) {
                    // (padding keys are counted too: they sit above every real key, and the rank looked for is below the real count)
                    const U k = key[i];
                    const bool match = (pass == 0) || ((k >> ((shift + 8) & (8 * PASSES - 1))) == (prefix >> ((shift + 8) & (8 * PASSES - 1))));
                    if (match) atomicAdd(&mine[(unsigned)((k >> shift) & 0xff)], 1u);
                }
                __syncthreads();
                {   // fold the copies (and leave them zero for the next pass): thread = (bin, quarter of the copies)
                    const int bin = tid & 255, part = tid >> 8;
                    unsigned sum = 0;
#pragma unroll
                    for (int c = 0; c < COPIES / 4; ++c) {
                        unsigned *w = hcopy + (part * (COPIES / 4) + c) * CSTRIDE + bin;
                        sum += *w;
                        *w = 0;
                    }
                    if (sum) atomicAdd(&hist[bin], sum);
                }
                __syncthreads();
            }
            if (tid < 64) {   // one wave scans the 256 bins: 4 bins per lane
                const unsigned *hsrc = counted ? hist0 : hist;
                unsigned h0 = hsrc[tid * 4], h1 = hsrc[tid * 4 + 1], h2 = hsrc[tid * 4 + 2], h3 = hsrc[tid * 4 + 3];
                if (!counted) {
                    if (pass == 0) { hist0[tid * 4] = h0; hist0[tid * 4 + 1] = h1; hist0[tid * 4 + 2] = h2; hist0[tid * 4 + 3] = h3; }
                    hist[tid * 4] = 0; hist[tid * 4 + 1] = 0; hist[tid * 4 + 2] = 0; hist[tid * 4 + 3] = 0;
                }
                unsigned tot = h0 + h1 + h2 + h3, incl = tot;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    unsigned o = __shfl_up(incl, off, 64);
                    if (tid >= off) incl += o;
                }
                const unsigned excl = incl - tot;
                if (rank >= excl && rank < incl) {   // exactly one lane
                    unsigned r = rank - excl;
                    int d;
                    if (r < h0) d = 0;
                    else if ((r -= h0) < h1) d = 1;
                    else if ((r -= h1) < h2) d = 2;
                    else { r -= h2; d = 3; }
                    s_prefix = prefix | ((U)(tid * 4 + d) << shift);
                    s_rank = r;
                }
            }
            __syncthreads();
            prefix = s_prefix;
            rank = s_rank;
        }
        const T vlo = Key<T>::dec(prefix);
        // next order statistic
        if (tid == 0) { s_cnt_le = 0; s_min_gt = ~0ull; }
        __syncthreads();
        unsigned cnt = 0;
        unsigned long long mg = ~0ull;
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const U k = key[i];   // a padding key can only be counted when the answer is NaN anyway, or be the minimum when it is not used
            if (k <= prefix) ++cnt;
            else if ((unsigned long long)k < mg) mg = (unsigned long long)k;
        }
        cnt = wave_sum(cnt);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(mg, off, 64);
            mg = o < mg ? o : mg;
        }
        if ((tid & 63) == 0) { atomicAdd(&s_cnt_le, cnt); atomicMin(&s_min_gt, mg); }
        __syncthreads();
        if (tid == 0) {
#pragma clang fp contract(off)   // hipcc would fuse the mul/add below even through the *_rn intrinsics
            T vhi = vlo;
            if (lo + 1 < nvalid && s_cnt_le < (unsigned)(lo + 2)) vhi = Key<T>::dec((U)s_min_gt);
            // numpy _lerp, without fma contraction so the rounding matches numpy's mul-then-add
            const double a = (double)vlo, b = (double)vhi, diff = __dsub_rn(b, a);
            double r = __dadd_rn(a, __dmul_rn(diff, t));
            if (t >= 0.5) r = __dsub_rn(b, __dmul_rn(diff, __dsub_rn(1.0, t)));
            if (t == 0.0) r = a;
            out[(int64_t)qi * G + g] = r;
        }
        __syncthreads();
    }
}

}  // namespace vcy

using namespace vcy;

extern "C" size_t vcy_fit_workspace_bytes(int64_t G) { return (size_t)FIT_CB * FIT_NMOM * (size_t)G * sizeof(double); }

extern "C" int vcy_fit_slope_moments(const void *Y, const void *X, double *moments, void *workspace, int64_t C, int64_t G,
                                     int64_t ld, int dtype, vcy_stream stream)
{
    VCY_REQUIRE(Y && X && moments && workspace, "fit_slope_moments: null pointer");
    VCY_REQUIRE(C > 0 && G > 0 && ld >= G, "fit_slope_moments: bad shape");
    VCY_REQUIRE(dtype == VCY_F32 || dtype == VCY_F64, "fit_slope_moments: bad dtype");
    const int N = dtype == VCY_F32 ? 4 : 2;
    VCY_REQUIRE(ld % N == 0, "fit_slope_moments: ld must keep rows 16-byte aligned");
    hipStream_t st = as_stream(stream);
    const int nvec = (int)((G + N - 1) / N);
    dim3 grid((unsigned)((nvec + 255) / 256), FIT_CB);
    if (dtype == VCY_F32) hipLaunchKernelGGL(k_moments_plain<float>, grid, dim3(256), 0, st, (const float *)Y, (const float *)X, (double *)workspace, (int)C, (int)G, ld);
    else hipLaunchKernelGGL(k_moments_plain<double>, grid, dim3(256), 0, st, (const double *)Y, (const double *)X, (double *)workspace, (int)C, (int)G, ld);
    VCY_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_fit_slope_reduce, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, st, (const double *)workspace, moments, (int)G);
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}

extern "C" int vcy_fit_slope_from_moments(const double *moments, float *gamma, int64_t G, vcy_stream stream)
{
    VCY_REQUIRE(moments && gamma && G > 0, "fit_slope_from_moments: bad arguments");
    hipLaunchKernelGGL(k_fit_slope_final, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, as_stream(stream), moments, gamma, (int)G);
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}

extern "C" int vcy_fit_slope(const void *Y, const void *X, float *gamma, void *workspace, int64_t C, int64_t G, int64_t ld,
                             int dtype, vcy_stream stream)
{
    VCY_REQUIRE(Y && X && gamma && workspace, "fit_slope: null pointer");
    // partials use FIT_CB*3*G doubles of the FIT_CB*FIT_NMOM*G workspace; the reduced moments sit behind them
    double *mom = (double *)workspace + (size_t)FIT_CB * 3 * (size_t)G;
    int rc = vcy_fit_slope_moments(Y, X, mom, workspace, C, G, ld, dtype, stream);
    if (rc) return rc;
    return vcy_fit_slope_from_moments(mom, gamma, G, stream);
}

extern "C" int vcy_fit_weighted(const void *Y, const void *X, int weight_mode, const void *W, const void *M, const void *M2,
                                const double *scale_a, const double *scale_b, const double *down, const double *up,
                                int fit_offset, int box_q, double lo_gamma, double up_gamma_default, const double *up_gamma,
                                const double *q_fixed, float *gamma, float *q, float *R2, void *workspace, int64_t C, int64_t G,
                                int64_t ld, int dtype, vcy_stream stream)
{
    VCY_REQUIRE(Y && X && gamma && workspace, "fit_weighted: null pointer");
    VCY_REQUIRE(C > 0 && G > 0 && ld >= G, "fit_weighted: bad shape");
    VCY_REQUIRE(dtype == VCY_F32 || dtype == VCY_F64, "fit_weighted: bad dtype");
    VCY_REQUIRE(weight_mode >= 0 && weight_mode <= 2, "fit_weighted: bad weight_mode");
    VCY_REQUIRE(weight_mode != 0 || W, "fit_weighted: weight_mode 0 needs W");
    VCY_REQUIRE(weight_mode != 1 || (M && down && up), "fit_weighted: weight_mode 1 needs M, down, up");
    VCY_REQUIRE(weight_mode != 1 || ((scale_a == nullptr) == (scale_b == nullptr) && (scale_a == nullptr || M2)), "fit_weighted: scale_a/scale_b/M2 go together");
    hipStream_t st = as_stream(stream);
    dim3 grid((unsigned)((G + 255) / 256), FIT_CB);
#define VCY_LAUNCH_W(T, MODE)                                                                                                  \
    hipLaunchKernelGGL((k_moments_weighted<T, MODE>), grid, dim3(256), 0, st, (const T *)Y, (const T *)X, (const T *)W, (const T *)M, \
                       (const T *)M2, scale_a, scale_b, down, up, (double *)workspace, (int)C, (int)G, ld)
    if (dtype == VCY_F32) {
        if (weight_mode == 0) VCY_LAUNCH_W(float, 0); else if (weight_mode == 1) VCY_LAUNCH_W(float, 1); else VCY_LAUNCH_W(float, 2);
    } else {
        if (weight_mode == 0) VCY_LAUNCH_W(double, 0); else if (weight_mode == 1) VCY_LAUNCH_W(double, 1); else VCY_LAUNCH_W(double, 2);
    }
#undef VCY_LAUNCH_W
    VCY_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_fit_weighted_final, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, st, (const double *)workspace, fit_offset,
                       box_q, lo_gamma, up_gamma_default, up_gamma, q_fixed, gamma, q, R2, (int)C, (int)G);
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}

// (5, G) fp64 per-gene raw moments over cells [sum x, sum y, sum xx, sum xy, sum yy]: the ingredients of the paired
// row correlation of filter_genes_by_phase_portrait (analysis.py:1285-1288) and of any unweighted per-gene fit.
__global__ void k_gene_moments_reduce(const double *__restrict__ part, double *__restrict__ mom, int G)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    for (int k = 0; k < 5; ++k) {
        double s = 0.0;
        for (int cb = 0; cb < FIT_CB; ++cb) s += part[((int64_t)cb * FIT_NMOM + k) * G + g];
        mom[(int64_t)k * G + g] = s;
    }
}

extern "C" int vcy_gene_moments(const void *Y, const void *X, double *moments, void *workspace, int64_t C, int64_t G, int64_t ld,
                                int dtype, vcy_stream stream)
{
    VCY_REQUIRE(Y && X && moments && workspace && C > 0 && G > 0 && ld >= G, "gene_moments: bad arguments");
    VCY_REQUIRE(dtype == VCY_F32 || dtype == VCY_F64, "gene_moments: bad dtype");
    hipStream_t st = as_stream(stream);
    dim3 grid((unsigned)((G + 255) / 256), FIT_CB);
    if (dtype == VCY_F32)
        hipLaunchKernelGGL((k_moments_weighted<float, 2>), grid, dim3(256), 0, st, (const float *)Y, (const float *)X, (const float *)nullptr, (const float *)nullptr,
                           (const float *)nullptr, (const double *)nullptr, (const double *)nullptr, (const double *)nullptr, (const double *)nullptr, (double *)workspace, (int)C, (int)G, ld);
    else
        hipLaunchKernelGGL((k_moments_weighted<double, 2>), grid, dim3(256), 0, st, (const double *)Y, (const double *)X, (const double *)nullptr, (const double *)nullptr,
                           (const double *)nullptr, (const double *)nullptr, (const double *)nullptr, (const double *)nullptr, (const double *)nullptr, (double *)workspace, (int)C, (int)G, ld);
    VCY_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_gene_moments_reduce, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, st, (const double *)workspace, moments, (int)G);
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}

extern "C" size_t vcy_quantile_workspace_bytes(int64_t C, int64_t G) { return (size_t)C * (size_t)G * sizeof(double); }

extern "C" int vcy_gene_quantiles(const void *M, const void *M2, const double *scale_a, const double *scale_b, const void *mask_src,
                                  const double *mask_thr, int mask_mode, const double *qs_host, int nq, double *out, void *workspace,
                                  int64_t C, int64_t G, int64_t ld, int dtype, vcy_stream stream)
{
    VCY_REQUIRE(mask_mode >= 0 && mask_mode <= 2 && (mask_mode == 0 || (mask_src && mask_thr)), "gene_quantiles: bad mask arguments");
    VCY_REQUIRE(M && qs_host && out && workspace && nq > 0 && nq <= 16, "gene_quantiles: bad arguments");
    VCY_REQUIRE(C > 0 && G > 0 && ld >= G, "gene_quantiles: bad shape");
    VCY_REQUIRE((scale_a == nullptr) == (scale_b == nullptr) && (scale_a == nullptr || M2), "gene_quantiles: scale_a/scale_b/M2 go together");
    VCY_REQUIRE(dtype == VCY_F32 || dtype == VCY_F64, "gene_quantiles: bad dtype");
    hipStream_t st = as_stream(stream);
    QArgs qs_dev;
    for (int i = 0; i < 16; ++i) qs_dev.q[i] = i < nq ? qs_host[i] : 0.0;
    for (int i = 0; i < nq; ++i) VCY_REQUIRE(qs_host[i] >= 0.0 && qs_host[i] <= 100.0, "gene_quantiles: percentile outside [0,100]");
    void *Z = workspace;   // gene-major key matrix (G, C) of dtype
    dim3 gridz((unsigned)((G + 63) / 64), (unsigned)((C + 63) / 64));
    if (dtype == VCY_F32) {
        hipLaunchKernelGGL(k_build_z<float>, gridz, dim3(256), 0, st, (const float *)M, (const float *)M2, scale_a, scale_b, (const float *)mask_src, mask_thr, mask_mode, (float *)Z, (int)C, (int)G, ld);
        VCY_LAUNCH_CHECK();
        // the gene's keys in the registers of 1024 threads when they fit (C <= 65 536 in f32), else the row is re-read per pass
        const bool reg = env_int("VCY_QUANTILES_REG", 1) == 1;
#define VCY_QREG(TT, NPT) hipLaunchKernelGGL((k_gene_quantiles_reg<TT, NPT>), dim3((unsigned)G), dim3(1024), 0, st, (const TT *)Z, qs_dev, nq, out, (int)C, (int)G, mask_mode != 0)
        if (reg && C <= 1024 * 16) VCY_QREG(float, 16);
        else if (reg && C <= 1024 * 32) VCY_QREG(float, 32);
        else if (reg && C <= 1024 * 48) VCY_QREG(float, 48);
        else if (reg && C <= 1024 * 56) VCY_QREG(float, 56);
        else if (reg && C <= 1024 * 64) VCY_QREG(float, 64);
        else hipLaunchKernelGGL(k_gene_quantiles<float>, dim3((unsigned)G), dim3(256), 0, st, (const float *)Z, qs_dev, nq, out, (int)C, (int)G, mask_mode != 0);
    } else {
        hipLaunchKernelGGL(k_build_z<double>, gridz, dim3(256), 0, st, (const double *)M, (const double *)M2, scale_a, scale_b, (const double *)mask_src, mask_thr, mask_mode, (double *)Z, (int)C, (int)G, ld);
        VCY_LAUNCH_CHECK();
        const bool reg = env_int("VCY_QUANTILES_REG", 1) == 1;          // f64 keys take two registers each: up to 32 768 cells
        if (reg && C <= 1024 * 8) VCY_QREG(double, 8);
        else if (reg && C <= 1024 * 16) VCY_QREG(double, 16);
        else if (reg && C <= 1024 * 24) VCY_QREG(double, 24);
        else if (reg && C <= 1024 * 32) VCY_QREG(double, 32);
        else if (reg && C <= 1024 * 40) VCY_QREG(double, 40);
        else if (reg && C <= 1024 * 50) VCY_QREG(double, 50);          // 100 of the 128 VGPRs a 1024-thread workgroup may use hold keys
        else hipLaunchKernelGGL(k_gene_quantiles<double>, dim3((unsigned)G), dim3(256), 0, st, (const double *)Z, qs_dev, nq, out, (int)C, (int)G, mask_mode != 0);
    }
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}
