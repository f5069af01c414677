// epsilon-SVR with an RBF kernel on scalar inputs: the noise model of score_cv_vs_mean (analysis.py:280-282, 324-326,
// sklearn.svm.SVR(gamma=150/G) on log2 mean -> log2 CV, one point per gene) and of adjust_totS_totU
// (analysis.py:844-851, SVR(C=100, gamma=1e-6) on the per-cell totals, one point per cell).
//
// scikit-learn (not vendored by the reference) hands these to libsvm 3.x: SMO on the 2n-variable dual
//     min 1/2 b'Qb + p'b,   y'b = 0,   0 <= b <= C,      b = [alpha; alpha*],  y = [+1; -1],  p = [eps - t; eps + t]
// with the second-order working-set selection of Fan, Chen & Lin (JMLR 2005, "WSS 2"), stopping when the maximal KKT
// violation m(b) - M(b) < tol (1e-3).  This file restates that published algorithm for the device:
//   * the gradient pair (G_k, G_{k+n}) is carried as ONE residual r_k = t_k - f(x_k):  -y G = r - eps (alpha part),
//     r + eps (alpha* part), so a sweep reads 17 bytes per point (x, r, bound flags) instead of libsvm's 2 x (G, alpha, y);
//   * kernel rows are never cached: K(x_i, x_k) = exp(-gamma (x_i - x_k)^2) is recomputed in fp64 (libsvm rounds its rows
//     to float; fp64 rows make the gradient exact and the result agree with libsvm to the solver tolerance);
//   * one SMO step = two sweeps over the points (apply the previous step + select i; select j), spread over up to 64
//     co-resident workgroups (cooperative launch) that meet at two grid barriers per step; every workgroup reduces the
//     per-workgroup winners redundantly, so no workgroup ever waits for a broadcast.  n <= 512 points run on one
//     workgroup with plain __syncthreads;
//   * up to 8 points per thread (131 072 points on 64 workgroups) live in registers for the whole solve (k_svr_smo<E>);
//     larger problems keep the same state in global memory (L2).
// A grid barrier that is not met within ~2 s sets the `failed` flag and every workgroup leaves (the host reports it): the
// kernel cannot hang the device.
#include "common.h"
#include <stdlib.h>

// the register-resident and the global-memory variants of the solver must round alike (same trajectory on any launch shape):
// no mul+add fusion beyond the explicit fma calls
#pragma clang fp contract(off)

namespace vcy {

constexpr int SVR_TPB = 256;
constexpr int SVR_MAX_WG = 64;
constexpr unsigned char SVR_A_LO = 1, SVR_A_HI = 2, SVR_S_LO = 4, SVR_S_HI = 8;     // alpha == 0, alpha == C, alpha* == 0, alpha* == C
constexpr double SVR_TAU = 1e-12;                                                    // libsvm's floor of the curvature

struct SvrSlot {          // one workgroup's winner of a sweep
    double v;             // -y G of the winner (select i) or the second-order objective (select j)
    double vmin;          // min over I_low of -y G (select i only)
    double x, r, beta;    // the winner's input, residual and dual variable
    long long idx;        // variable index in [0, 2n), -1 when the workgroup has no candidate
};

struct SvrShared {        // global scratch shared by the workgroups of one fit
    unsigned bar;
    int failed;
    int pad[2];
    SvrSlot slot_i[SVR_MAX_WG];
    SvrSlot slot_j[SVR_MAX_WG];
};

template <typename T> __device__ __forceinline__ void st_agent(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T> __device__ __forceinline__ T ld_agent(T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// All workgroups of the grid meet here.  `epoch` counts the barriers this thread has passed.  Returns false when the
// other workgroups did not arrive (or someone already failed).
__device__ __forceinline__ bool svr_grid_barrier(SvrShared *sh, int nwg, unsigned &epoch, int *lds_flag)
{
    ++epoch;
    if (nwg == 1) { __syncthreads(); return true; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int ok = 1;
        __hip_atomic_fetch_add(&sh->bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = epoch * (unsigned)nwg;
        long long spins = 0;
        while (__hip_atomic_load(&sh->bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 1023) == 0 && (spins > (4ll << 20) || ld_agent(&sh->failed))) { ok = 0; st_agent(&sh->failed, 1); break; }
        }
        *lds_flag = ok;
    }
    __syncthreads();
    return *lds_flag != 0;
}

struct SvrCand { double v; long long idx; };
__device__ __forceinline__ SvrCand better_max(SvrCand a, SvrCand b) { return (b.v > a.v || (b.v == a.v && b.idx >= 0 && (a.idx < 0 || b.idx < a.idx))) ? b : a; }
__device__ __forceinline__ SvrCand better_min(SvrCand a, SvrCand b) { return (b.v < a.v || (b.v == a.v && b.idx >= 0 && (a.idx < 0 || b.idx < a.idx))) ? b : a; }

template <bool MAX> __device__ __forceinline__ SvrCand wave_best(SvrCand c)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        SvrCand o;
        o.v = __shfl_xor(c.v, off);
        o.idx = __shfl_xor(c.idx, off);
        c = MAX ? better_max(c, o) : better_min(c, o);
    }
    return c;
}
__device__ __forceinline__ double wave_min_d(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off));
    return v;
}

__device__ __forceinline__ unsigned char svr_flags(unsigned char st, int part, double beta, double Cbox)
{
    if (part == 0) { st &= ~(SVR_A_LO | SVR_A_HI); if (beta <= 0.0) st |= SVR_A_LO; if (beta >= Cbox) st |= SVR_A_HI; }
    else           { st &= ~(SVR_S_LO | SVR_S_HI); if (beta <= 0.0) st |= SVR_S_LO; if (beta >= Cbox) st |= SVR_S_HI; }
    return st;
}

// info: [0] SMO steps taken, [1] converged, [2] a grid barrier failed, [3] workgroups used.
// E > 0: a thread owns at most E points (k0 + e * step) and keeps their x, r, alpha, alpha* and bound flags in registers for the
// whole solve (n <= gridDim.x * 256 * E), so a sweep touches no memory at all; E == 0: any n, state in global memory (L2).
// Both run the same arithmetic in the same order: the trajectory is the same.
template <int E>
__global__ __launch_bounds__(SVR_TPB) void k_svr_smo(const double *__restrict__ x, const double *__restrict__ t, double *__restrict__ r,
                                                      double *__restrict__ alpha /* (2, n) */, unsigned char *__restrict__ status,
                                                      SvrShared *__restrict__ sh, int *__restrict__ info, int n, double Cbox, double eps,
                                                      double gamma, double tol, long long max_iter)
{
    __shared__ SvrCand s_c[SVR_TPB / VCY_WAVE];
    __shared__ double s_m[SVR_TPB / VCY_WAVE];
    __shared__ SvrSlot s_win;
    __shared__ double s_vmin;
    __shared__ int s_flag;
    const int nwg = gridDim.x, wg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int step = nwg * SVR_TPB, k0 = wg * SVR_TPB + tid;
    unsigned epoch = 0;

    constexpr bool REG = E > 0;
    constexpr int EE = REG ? E : 1;
    double xs[EE], rs[EE], a0[EE], a1[EE];
    unsigned char ss[EE];
#define SVR_FOR_OWNED(e, k) for (int e = 0, k = k0; (!REG || e < E) && k < n; ++e, k += step)
#pragma unroll
    SVR_FOR_OWNED(e, k) {                          // b = 0: f = 0, r = t, every variable at its lower bound
        if (REG) { xs[e] = x[k]; rs[e] = t[k]; a0[e] = 0.0; a1[e] = 0.0; ss[e] = SVR_A_LO | SVR_S_LO; }
        else {
            r[k] = t[k];
            alpha[k] = 0.0;
            alpha[(size_t)n + k] = 0.0;
            status[k] = SVR_A_LO | SVR_S_LO;
        }
    }
    bool pending = false, ok = true, converged = false;
    int pi = -1, pj = -1, parti = 0, partj = 0;
    double xi = 0.0, xj = 0.0, dci = 0.0, dcj = 0.0, bi_new = 0.0, bj_new = 0.0;
    long long it = 0;

    // every workgroup reduces the nwg published winners (all fields fetched at once: one memory latency); maxsel: largest v
    // (select i, also folds vmin), else smallest v
    auto gather = [&](SvrSlot *slots, bool maxsel) {
        if (wv == 0) {
            SvrCand c{maxsel ? -INFINITY : INFINITY, -1};
            double vm = INFINITY, px = 0.0, pr = 0.0, pb = 0.0;
            if (lane < nwg) {
                c.v = ld_agent(&slots[lane].v);
                c.idx = ld_agent(&slots[lane].idx);
                vm = ld_agent(&slots[lane].vmin);
                px = ld_agent(&slots[lane].x);
                pr = ld_agent(&slots[lane].r);
                pb = ld_agent(&slots[lane].beta);
                if (c.idx < 0) c.v = maxsel ? -INFINITY : INFINITY;
            }
            const SvrCand w = maxsel ? wave_best<true>(c) : wave_best<false>(c);
            vm = wave_min_d(vm);
            const unsigned long long m = __ballot(lane < nwg && c.idx == w.idx && w.idx >= 0);
            const int src = m ? __ffsll((long long)m) - 1 : 0;
            px = __shfl(px, src); pr = __shfl(pr, src); pb = __shfl(pb, src);
            if (lane == 0) { s_win.v = w.v; s_win.idx = w.idx; s_win.x = px; s_win.r = pr; s_win.beta = pb; s_vmin = vm; }
        }
        __syncthreads();
    };
    // the thread that holds the workgroup's winner publishes it with its payload; thread 0 adds vmin (and "no candidate")
    auto publish = [&](SvrSlot *slots, SvrCand w, SvrCand mine, double mx, double mr, double mb, double vmin) {
        if (w.idx >= 0 && mine.idx == w.idx) {
            st_agent(&slots[wg].v, w.v);
            st_agent(&slots[wg].x, mx);
            st_agent(&slots[wg].r, mr);
            st_agent(&slots[wg].beta, mb);
            st_agent(&slots[wg].idx, w.idx);
        }
        if (tid == 0) {
            st_agent(&slots[wg].vmin, vmin);
            if (w.idx < 0) st_agent(&slots[wg].idx, (long long)-1);
        }
    };

    while (true) {
        // ---- sweep 1: apply the previous step to the residuals, then i = argmax over I_up of -y G and min over I_low
        SvrCand up{-INFINITY, -1};
        double low = INFINITY, ux = 0.0, ur = 0.0, ub = 0.0;
#pragma unroll
        SVR_FOR_OWNED(e, k) {
            const double xk = REG ? xs[e] : x[k];
            double rk = REG ? rs[e] : r[k];
            unsigned char st = REG ? ss[e] : status[k];
            if (pending) {
                const double di = xk - xi, dj = xk - xj;
                rk -= dci * exp(-gamma * di * di) + dcj * exp(-gamma * dj * dj);
                if (k == pi) st = svr_flags(st, parti, bi_new, Cbox);
                if (k == pj) st = svr_flags(st, partj, bj_new, Cbox);
                if (REG) {
                    rs[e] = rk; ss[e] = st;
                    if (k == pi) { if (parti == 0) a0[e] = bi_new; else a1[e] = bi_new; }
                    if (k == pj) { if (partj == 0) a0[e] = bj_new; else a1[e] = bj_new; }
                } else {
                    r[k] = rk;
                    if (k == pi) { alpha[(size_t)parti * n + k] = bi_new; status[k] = st; }
                    if (k == pj) { alpha[(size_t)partj * n + k] = bj_new; status[k] = st; }
                }
            }
            const double vA = rk - eps, vS = rk + eps;
            const long long before = up.idx;
            if (!(st & SVR_A_HI)) up = better_max(up, SvrCand{vA, (long long)k});
            if (!(st & SVR_S_LO)) up = better_max(up, SvrCand{vS, (long long)k + n});
            if (up.idx != before) { ux = xk; ur = rk; if (REG) ub = up.idx >= n ? a1[e] : a0[e]; }
            if (!(st & SVR_A_LO)) low = fmin(low, vA);
            if (!(st & SVR_S_HI)) low = fmin(low, vS);
        }
        {
            const SvrCand mine = up;
            if (!REG) ub = mine.idx >= 0 ? alpha[mine.idx] : 0.0;              // alpha is (2, n): variable idx lives at alpha[idx]
            up = wave_best<true>(up);
            low = wave_min_d(low);
            if (lane == 0) { s_c[wv] = up; s_m[wv] = low; }
            __syncthreads();
            up = s_c[0]; low = s_m[0];
            for (int w = 1; w < SVR_TPB / VCY_WAVE; ++w) { up = better_max(up, s_c[w]); low = fmin(low, s_m[w]); }
            publish(sh->slot_i, up, mine, ux, ur, ub, low);
        }
        if (!(ok = svr_grid_barrier(sh, nwg, epoch, &s_flag))) break;
        gather(sh->slot_i, true);
        const long long i = s_win.idx;
        const double Gmax = s_win.v, Gmin = s_vmin, ri = s_win.r, bi = s_win.beta;
        xi = s_win.x;
        __syncthreads();                                   // s_win is rewritten by the second gather
        if (i < 0 || !(Gmax - Gmin >= tol)) { converged = true; break; }
        if (it >= max_iter) break;

        // ---- sweep 2: j = argmin over I_low, -y G < Gmax, of -(Gmax + y G)^2 / (K_ii + K_jj - 2 K_ij)
        SvrCand best{INFINITY, -1};
        double jx = 0.0, jr = 0.0, jb = 0.0;
#pragma unroll
        SVR_FOR_OWNED(e, k) {
            const double xk = REG ? xs[e] : x[k], d = xk - xi, rk = REG ? rs[e] : r[k];
            const unsigned char st = REG ? ss[e] : status[k];
            double a = 2.0 - 2.0 * exp(-gamma * d * d);
            if (!(a > 0.0)) a = SVR_TAU;
            const double bA = Gmax - (rk - eps), bS = Gmax - (rk + eps), ninv = -1.0 / a;
            const long long before = best.idx;
            if (!(st & SVR_A_LO) && bA > 0.0) best = better_min(best, SvrCand{bA * bA * ninv, (long long)k});
            if (!(st & SVR_S_HI) && bS > 0.0) best = better_min(best, SvrCand{bS * bS * ninv, (long long)k + n});
            if (best.idx != before) { jx = xk; jr = rk; if (REG) jb = best.idx >= n ? a1[e] : a0[e]; }
        }
        {
            const SvrCand mine = best;
            if (!REG) jb = mine.idx >= 0 ? alpha[mine.idx] : 0.0;
            best = wave_best<false>(best);
            if (lane == 0) s_c[wv] = best;
            __syncthreads();
            best = s_c[0];
            for (int w = 1; w < SVR_TPB / VCY_WAVE; ++w) best = better_min(best, s_c[w]);
            publish(sh->slot_j, best, mine, jx, jr, jb, 0.0);
        }
        if (!(ok = svr_grid_barrier(sh, nwg, epoch, &s_flag))) break;
        gather(sh->slot_j, false);
        const long long j = s_win.idx;
        const double rj = s_win.r, bj = s_win.beta;
        xj = s_win.x;
        __syncthreads();
        if (j < 0) { converged = true; break; }

        // ---- the two-variable subproblem (Platt's analytic step with libsvm's clipping order); every thread computes it
        pi = (int)(i % n); parti = (int)(i / n);
        pj = (int)(j % n); partj = (int)(j / n);
        const double Gi = parti == 0 ? eps - ri : ri + eps;              // G_k = eps - r_k, G_{k+n} = eps + r_k
        const double Gj = partj == 0 ? eps - rj : rj + eps;
        const double dij = xi - xj, Kij = exp(-gamma * dij * dij);
        double ai = bi, aj = bj;
        if (parti != partj) {                                            // y_i != y_j: Q_ij = -K_ij
            double quad = 2.0 - 2.0 * Kij;
            if (!(quad > 0.0)) quad = SVR_TAU;
            const double delta = (-Gi - Gj) / quad, diff = ai - aj;
            ai += delta; aj += delta;
            if (diff > 0.0) { if (aj < 0.0) { aj = 0.0; ai = diff; } }
            else            { if (ai < 0.0) { ai = 0.0; aj = -diff; } }
            if (diff > 0.0) { if (ai > Cbox) { ai = Cbox; aj = Cbox - diff; } }
            else            { if (aj > Cbox) { aj = Cbox; ai = Cbox + diff; } }
        } else {
            double quad = 2.0 - 2.0 * Kij;
            if (!(quad > 0.0)) quad = SVR_TAU;
            const double delta = (Gi - Gj) / quad, sum = ai + aj;
            ai -= delta; aj += delta;
            if (sum > Cbox) { if (ai > Cbox) { ai = Cbox; aj = sum - Cbox; } }
            else            { if (aj < 0.0) { aj = 0.0; ai = sum; } }
            if (sum > Cbox) { if (aj > Cbox) { aj = Cbox; ai = sum - Cbox; } }
            else            { if (ai < 0.0) { ai = 0.0; aj = sum; } }
        }
        bi_new = ai; bj_new = aj;
        dci = (parti == 0 ? 1.0 : -1.0) * (ai - bi);                     // change of coef = alpha - alpha* at the two points
        dcj = (partj == 0 ? 1.0 : -1.0) * (aj - bj);
        pending = true;
        ++it;
    }
    if (REG) {                                        // hand the state to k_svr_finish
#pragma unroll
        SVR_FOR_OWNED(e, k) { r[k] = rs[e]; alpha[k] = a0[e]; alpha[(size_t)n + k] = a1[e]; status[k] = ss[e]; }
    }
#undef SVR_FOR_OWNED
    if (wg == 0 && tid == 0) {
        info[0] = (int)(it > 0x7fffffffll ? 0x7fffffffll : it);
        info[1] = converged ? 1 : 0;
        info[2] = ok ? 0 : 1;
        info[3] = nwg;
    }
}

// rho as libsvm's calculate_rho (mean of y G over the free variables, else the midpoint of the bounds) and the dual
// coefficients coef = alpha - alpha*.  One workgroup; out: coef (n), intercept (1) = -rho.
__global__ __launch_bounds__(1024) void k_svr_finish(const double *__restrict__ r, const double *__restrict__ alpha,
                                                      const unsigned char *__restrict__ status, double *__restrict__ coef,
                                                      double *__restrict__ intercept, int n, double eps)
{
    __shared__ double s_sum[16], s_ub[16], s_lb[16];
    __shared__ long long s_nf[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double sum = 0.0, ub = INFINITY, lb = -INFINITY;
    long long nf = 0;
    for (int k = tid; k < n; k += 1024) {
        const double rk = r[k];
        const unsigned char st = status[k];
        coef[k] = alpha[k] - alpha[(size_t)n + k];
        const double yA = eps - rk, yS = -(rk + eps);                     // y G of the alpha and alpha* variables
        if (st & SVR_A_HI) lb = fmax(lb, yA); else if (st & SVR_A_LO) ub = fmin(ub, yA); else { ++nf; sum += yA; }
        if (st & SVR_S_HI) ub = fmin(ub, yS); else if (st & SVR_S_LO) lb = fmax(lb, yS); else { ++nf; sum += yS; }
    }
    sum = wave_sum(sum);
    nf = (long long)wave_sum((double)nf);
    ub = wave_min_d(ub);
    lb = -wave_min_d(-lb);
    if (lane == 0) { s_sum[wv] = sum; s_nf[wv] = nf; s_ub[wv] = ub; s_lb[wv] = lb; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w) { sum += s_sum[w]; nf += s_nf[w]; ub = fmin(ub, s_ub[w]); lb = fmax(lb, s_lb[w]); }
        const double rho = nf > 0 ? sum / (double)nf : 0.5 * (ub + lb);
        *intercept = -rho;
    }
}

// decision function: out[q] = sum_k coef_k exp(-gamma (xq - x_k)^2) + intercept.  thread = one query, support points
// staged through LDS 1024 at a time (zero coefficients skipped per tile entry).
__global__ __launch_bounds__(256) void k_svr_predict(const double *__restrict__ x, const double *__restrict__ coef, const double *__restrict__ intercept,
                                                      const double *__restrict__ xq, double *__restrict__ out, int n, int m, double gamma)
{
    __shared__ double s_x[1024], s_c[1024];
    const int q = blockIdx.x * 256 + threadIdx.x;
    const double v = q < m ? xq[q] : 0.0;
    double acc0 = 0.0, acc1 = 0.0;
    for (int base = 0; base < n; base += 1024) {
        const int cnt = min(1024, n - base);
        __syncthreads();
        for (int u = threadIdx.x; u < 1024; u += 256) {
            s_x[u] = u < cnt ? x[base + u] : 0.0;
            s_c[u] = u < cnt ? coef[base + u] : 0.0;
        }
        __syncthreads();
        for (int u = 0; u < cnt; u += 2) {
            const double c0 = s_c[u], c1 = s_c[u + 1];
            if (c0 != 0.0) { const double d = v - s_x[u]; acc0 = fma(c0, exp(-gamma * d * d), acc0); }
            if (c1 != 0.0) { const double d = v - s_x[u + 1]; acc1 = fma(c1, exp(-gamma * d * d), acc1); }
        }
    }
    if (q < m) out[q] = acc0 + acc1 + *intercept;
}

}  // namespace vcy

using namespace vcy;

extern "C" int64_t vcy_svr_workspace_bytes(int64_t n)
{
    if (n < 0) return 0;
    // r (n) + alpha (2n) fp64, shared block, status bytes
    return (int64_t)sizeof(double) * 3 * n + (int64_t)sizeof(SvrShared) + ((n + 255) / 256) * 256 + 256;
}

extern "C" int vcy_svr_rbf_fit(const double *x, const double *t, double *coef, double *intercept, int32_t *info, void *workspace,
                               int64_t n, double C, double epsilon, double gamma, double tol, int64_t max_iter, vcy_stream stream)
{
    VCY_REQUIRE(x && t && coef && intercept && info && workspace, "svr_rbf_fit: null pointer");
    VCY_REQUIRE(n >= 1 && n < (1ll << 30), "svr_rbf_fit: n out of range");
    VCY_REQUIRE(C > 0.0 && epsilon >= 0.0 && gamma >= 0.0 && tol > 0.0, "svr_rbf_fit: C, tol must be > 0 and epsilon, gamma >= 0");
    hipStream_t s = (hipStream_t)stream;
    char *w = (char *)workspace;
    double *r = (double *)w;
    double *alpha = r + n;
    SvrShared *sh = (SvrShared *)(alpha + 2 * n);
    unsigned char *status = (unsigned char *)(sh + 1);
    if (max_iter <= 0) max_iter = n > 100000 ? 100 * n : 10000000;          // libsvm's own cap: max(10^7, 100 l)
    VCY_CHECK_HIP(hipMemsetAsync(sh, 0, sizeof(SvrShared), s));
    VCY_CHECK_HIP(hipMemsetAsync(info, 0, 4 * sizeof(int32_t), s));
    int nwg = (int)((n + 2 * SVR_TPB - 1) / (2 * SVR_TPB));
    if (nwg > SVR_MAX_WG) nwg = SVR_MAX_WG;
    if (const char *e = getenv("VCY_SVR_WG")) { int v = atoi(e); if (v >= 1 && v <= SVR_MAX_WG) nwg = v; }
    int ni = (int)n;
    long long mi = max_iter;
    void *args[] = {(void *)&x, (void *)&t, (void *)&r, (void *)&alpha, (void *)&status, (void *)&sh, (void *)&info,
                    (void *)&ni, (void *)&C, (void *)&epsilon, (void *)&gamma, (void *)&tol, (void *)&mi};
    auto kernel_for = [&](int wgs) -> const void * {         // points per thread -> register-resident variant, else the generic one
        const int64_t per = (n + (int64_t)wgs * SVR_TPB - 1) / ((int64_t)wgs * SVR_TPB);
        if (getenv("VCY_SVR_GLOBAL")) return (const void *)k_svr_smo<0>;
        return per <= 1 ? (const void *)k_svr_smo<1> : per <= 2 ? (const void *)k_svr_smo<2> : per <= 4 ? (const void *)k_svr_smo<4>
             : per <= 8 ? (const void *)k_svr_smo<8> : (const void *)k_svr_smo<0>;
    };
    if (nwg > 1) {
        // co-residency of the workgroups is what makes the grid barrier safe: ask the runtime for it
        hipError_t e = hipLaunchCooperativeKernel(kernel_for(nwg), dim3(nwg), dim3(SVR_TPB), args, 0, s);
        if (e != hipSuccess) { (void)hipGetLastError(); nwg = 1; }
    }
    if (nwg == 1) VCY_CHECK_HIP(hipLaunchKernel(kernel_for(1), dim3(1), dim3(SVR_TPB), args, 0, s));
    hipLaunchKernelGGL(k_svr_finish, dim3(1), dim3(1024), 0, s, r, alpha, status, coef, intercept, ni, epsilon);
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}

extern "C" int vcy_svr_rbf_predict(const double *x, const double *coef, const double *intercept, const double *xq, double *out,
                                   int64_t n, int64_t m, double gamma, vcy_stream stream)
{
    VCY_REQUIRE(x && coef && intercept && (m == 0 || (xq && out)), "svr_rbf_predict: null pointer");
    VCY_REQUIRE(n >= 1 && n < (1ll << 30) && m >= 0 && m < (1ll << 30), "svr_rbf_predict: sizes out of range");
    if (m == 0) return VCY_OK;
    hipLaunchKernelGGL(k_svr_predict, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, coef, intercept, xq, out,
                       (int)n, (int)m, gamma);
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}
