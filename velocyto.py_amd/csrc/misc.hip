// misc.hip -- the small streaming kernels around the four hot stages:
//   normalisation pre-step (analysis.py:535-631), the dmat transform of estimate_transition_prob
//   on a stored delta_S (analysis.py:1538, 1575-1601), the diag/NaN fix-ups (:1604-1612), the
//   transition-probability / embedding-shift step in neighbour-list form (:1670-1733) and the
//   Markov step of Diffusion.diffuse (diffusion.py:93-105).
#include <math.h>
#include "common.h"

namespace vcy {

// ---------------------------------------------------------------- a1: normalisation
// cell_size[c] = sum_g M[c,g]  (S.sum(0) in the reference's layout): one wave per cell row.
template <typename T>
__global__ __launch_bounds__(256) void k_row_sums(const T *__restrict__ M, double *__restrict__ out, int64_t C, int G, int64_t ld)
{
    using V = typename Vec<T>::type;
    constexpr int N = Vec<T>::N;
    const int lane = threadIdx.x & 63;
    const int64_t c = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (c >= C) return;
    const T *row = M + c * ld;
    double s = 0.0;
    const int nvec = G / N;
    for (int v = lane; v < nvec; v += 64) {
        const V x = reinterpret_cast<const V *>(row)[v];
        const T *xp = reinterpret_cast<const T *>(&x);
#pragma unroll
        for (int k = 0; k < N; ++k) s += (double)xp[k];
    }
    for (int g = nvec * N + lane; g < G; g += 64) s += (double)row[g];
    s = wave_sum(s);
    if (lane == 0) out[c] = s;
}

// out_sz[c,g] = factor[c] * M[c,g] (non-finite -> 0 when fix != 0, analysis.py:580);
// out_norm[c,g] = log2(out_sz + pcount) (analysis.py:551); either output may be NULL.
template <typename T>
__global__ __launch_bounds__(256) void k_scale_log(const T *__restrict__ M, const double *__restrict__ factor, T *__restrict__ out_sz,
                                                    T *__restrict__ out_norm, int64_t C, int G, int64_t ld, double pcount, int fix)
{
    const int64_t total = C * ld;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = t / ld;
        const int g = (int)(t - c * ld);
        T v = T(0), l = T(0);
        if (g < G) {
            double x = (factor ? factor[c] : 1.0) * (double)M[t];
            if (fix && !isfinite(x)) x = 0.0;
            v = (T)x;
            l = (T)log2(x + pcount);
        }
        if (out_sz) out_sz[t] = v;
        if (out_norm) out_norm[t] = l;
    }
}

// ---------------------------------------------------------------- dmat from a stored delta_S
template <typename T>
__global__ __launch_bounds__(256) void k_delta_transform(const T *__restrict__ hi, const T *__restrict__ dS, T *__restrict__ dmat,
                                                          T *__restrict__ e_out, int64_t C, int G, int64_t ld, T used_dt, int mode, T psc)
{
    // mode 0 linear, 1 sqrt, 2 log10 (sign(D) f(|D|+psc), D = (hi + dt*dS) - hi); 3 logratio:
    // e_out = log2(hi + psc), dmat = log2(|hi + dt*dS| + psc) - e_out   (analysis.py:1583-1584)
    const int64_t total = C * ld;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(t % ld);
        T d = T(0), eo = T(0);
        if (g < G) {
            const T h = hi[t];
            const T ht = h + used_dt * dS[t];
            if (mode == 3) {
                eo = (T)log2((double)(h + psc));
                d = (T)log2((double)(fabs(ht) + psc)) - eo;
            } else {
                const T D = ht - h;
                if (mode == 0) d = D;
                else {
                    const double a = (double)fabs(D) + (double)psc;
                    const T f = (T)(mode == 1 ? sqrt(a) : log10(a));
                    d = D > T(0) ? f : (D < T(0) ? -f : T(0) * f);
                }
            }
        }
        dmat[t] = d;
        if (e_out) e_out[t] = eo;
    }
}

// ---------------------------------------------------------------- randomised control: permute_rows_nsign (analysis.py:2407-2420)
// Per gene, the values shuffled across the cells and multiplied by random signs.  The reference draws both from numba's private
// Mersenne twister, gene after gene; here every gene gets its own pseudo-random PERMUTATION of [0, C) as a function that is evaluated
// where it is needed - a 5-round Feistel network over the 2 * hb >= log2(C) bits of the cell number, keyed by (seed, gene), walked
// until it lands below C (a permutation of [0, 4^hb) restricted to the cycle-walk over a subset is a permutation of the subset) - so
// the shuffle is one gather, out[c, g] = +- in[pi_g(c), g], with no sort, no keys in memory and no gene-major copy.  Statistical
// parity with the reference (uniform, independent per gene), not the same random numbers.
__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t perm_key(uint32_t seed_lo, uint32_t seed_hi, int g) { return mix32(seed_lo ^ mix32((uint32_t)g * 0x9e3779b9u + seed_hi)); }
__device__ __forceinline__ uint32_t perm_cell(uint32_t c, uint32_t key, int hb, uint32_t C)
{
    const uint32_t mask = (1u << hb) - 1u;
    uint32_t x = c;
    do {
        uint32_t L = x >> hb, R = x & mask;
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const uint32_t t = L ^ (mix32(R + key + (uint32_t)r * 0x85ebca6bu) & mask);
            L = R;
            R = t;
        }
        x = (L << hb) | R;
    } while (x >= C);
    return x;
}
__device__ __forceinline__ uint32_t perm_sign(uint32_t c, uint32_t key) { return mix32(key ^ (c * 0xc2b2ae35u + 0x27d4eb2fu)) >> 31; }

template <typename T, int CPT>
__global__ __launch_bounds__(256) void k_permute_rows_nsign(const T *__restrict__ in, T *__restrict__ out, int C, int G, int64_t ld,
                                                             uint32_t seed_lo, uint32_t seed_hi, int hb)
{
    const int g = blockIdx.x * 256 + threadIdx.x;            // lanes = consecutive genes: coalesced stores, one sector per gathered value
    if (g >= ld) return;
    const int c0 = blockIdx.y * CPT;
    if (g >= G) {                                             // padding columns of the cell-major layout stay zero
        for (int c = c0; c < min(C, c0 + CPT); ++c) out[(int64_t)c * ld + g] = T(0);
        return;
    }
    const uint32_t key = perm_key(seed_lo, seed_hi, g);
    T v[CPT];
    uint32_t sg[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = min(c0 + i, C - 1);
        v[i] = in[(int64_t)perm_cell((uint32_t)c, key, hb, (uint32_t)C) * ld + g];
        sg[i] = perm_sign((uint32_t)c, key);
    }
#pragma unroll
    for (int i = 0; i < CPT; ++i)
        if (c0 + i < C) out[(int64_t)(c0 + i) * ld + g] = sg[i] ? -v[i] : v[i];
}

// The same shuffle on a GENE-MAJOR copy (G, C): a gene's values are contiguous, so the gather stays inside one 4 C-byte row that
// lives in L2 - the cell-major gather above moves a 64-byte sector per 4-byte value (96 GB for a 6 GB matrix, 33 ms); two tiled
// transposes and this kernel move 36 GB.  Lanes = consecutive cells of one gene.
template <typename T, int CPT>
__global__ __launch_bounds__(256) void k_permute_within_gene_rows(const T *__restrict__ in, T *__restrict__ out, int C, uint32_t seed_lo,
                                                                   uint32_t seed_hi, int hb)
{
    const int g = blockIdx.y;
    const uint32_t key = perm_key(seed_lo, seed_hi, g);
    const T *row = in + (int64_t)g * C;
    T *orow = out + (int64_t)g * C;
    const int c0 = blockIdx.x * 256 * CPT + threadIdx.x;
    T v[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = min(c0 + i * 256, C - 1);
        const T x = row[perm_cell((uint32_t)c, key, hb, (uint32_t)C)];
        v[i] = perm_sign((uint32_t)c, key) ? -x : x;
    }
#pragma unroll
    for (int i = 0; i < CPT; ++i)
        if (c0 + i * 256 < C) orow[c0 + i * 256] = v[i];
}

// np.fill_diagonal(corrcoef, 0); corrcoef[isnan] = nan_to   (analysis.py:1604-1606) on the compact form
template <typename T>
__global__ void k_corr_fixup(T *__restrict__ vals, const int32_t *__restrict__ ixs, int64_t cell0, int64_t total, int nrndm,
                             int zero_self, int fix_nan, T nan_to, int *__restrict__ nan_count)
{
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = cell0 + t / nrndm;
        T v = vals[t];
        if (zero_self && ixs[t] == c) v = T(0);
        else if (v != v) {
            if (nan_count) atomicAdd(nan_count, 1);
            if (fix_nan) v = nan_to;
        }
        vals[t] = v;
    }
}

// ---------------------------------------------------------------- stage E: transition probabilities
// One wave per cell over its neighbour list (n entries):
//   p_n = exp(corr[c,n]/sigma) / sum_n exp(corr/sigma)                 (analysis.py:1697-1698)
//   u_n = (emb[i_n] - emb[c]) / |emb[i_n] - emb[c]|, 0 on the diagonal  (:1704-1708; 0/0 = NaN otherwise)
//   delta_embedding[c] = sum_n p_n u_n - (1/n) sum_n u_n               (:1710-1712)
//   wdiff[c,n] = p_n - 1/n        (weights of the expression-scaling pooling, :1716)
template <typename T>
__global__ __launch_bounds__(256) void k_transition_prob(const T *__restrict__ corr, const int32_t *__restrict__ ixs,
                                                          const double *__restrict__ emb, int edim, T *__restrict__ tp, T *__restrict__ wdiff,
                                                          double *__restrict__ delta_emb, int64_t cell0, int64_t C_out, int n, double sigma)
{
    const int lane = threadIdx.x & 63;
    const int64_t cl = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (cl >= C_out) return;
    const int64_t c = cell0 + cl;
    const T *crow = corr + cl * n;
    const int32_t *irow = ixs + cl * n;
    double z = 0.0;
    for (int k = lane; k < n; k += 64) z += exp((double)crow[k] / sigma);
    z = wave_sum(z);
    double acc[4] = {0, 0, 0, 0};   // edim <= 4
    for (int k = lane; k < n; k += 64) {
        const double p = exp((double)crow[k] / sigma) / z;
        const int i = irow[k];
        if (tp) tp[cl * n + k] = (T)p;
        if (wdiff) wdiff[cl * n + k] = (T)(p - 1.0 / n);
        double nrm = 0.0, dv[4];
        for (int a = 0; a < edim; ++a) { dv[a] = emb[(int64_t)i * edim + a] - emb[c * edim + a]; nrm += dv[a] * dv[a]; }
        nrm = sqrt(nrm);
        for (int a = 0; a < edim; ++a) {
            const double u = (i == c) ? 0.0 : dv[a] / nrm;
            acc[a] += (p - 1.0 / n) * u;
        }
    }
    for (int a = 0; a < edim; ++a) {
        const double s = wave_sum(acc[a]);
        if (lane == 0) delta_emb[cl * edim + a] = s;
    }
}

// cos_proj[c] = sum_g a[c,g] b[c,g] / sqrt(sum_g b[c,g]^2)   (analysis.py:1717), one wave per cell
template <typename T>
__global__ __launch_bounds__(256) void k_row_cosproj(const T *__restrict__ A, const T *__restrict__ B, double *__restrict__ out, int64_t C,
                                                      int G, int64_t ld)
{
    const int lane = threadIdx.x & 63;
    const int64_t c = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (c >= C) return;
    const T *a = A + c * ld, *b = B + c * ld;
    double sab = 0, sbb = 0;
    for (int g = lane; g < G; g += 64) { const double x = a[g], y = b[g]; sab = fma(x, y, sab); sbb = fma(y, y, sbb); }
    sab = wave_sum(sab); sbb = wave_sum(sbb);
    if (lane == 0) out[c] = sab / sqrt(sbb);
}

// ---------------------------------------------------------------- stage F: Markov step  y = x . T
// dense row-major T (n, n): y[j] = sum_i x[i] T[i,j]; lanes over j (coalesced rows), rows split over
// blockIdx.y with a deterministic two-stage reduction through `part` (gridDim.y, n).
template <typename T>
__global__ __launch_bounds__(256) void k_vecmat_dense(const T *__restrict__ Tm, const double *__restrict__ x, double *__restrict__ part, int n)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int per = (n + gridDim.y - 1) / gridDim.y;
    const int i0 = blockIdx.y * per, i1 = min(n, i0 + per);
    double acc = 0.0;
    for (int i = i0; i < i1; ++i) acc = fma(x[i], (double)Tm[(int64_t)i * n + j], acc);
    part[(int64_t)blockIdx.y * n + j] = acc;
}
// the same sums with 16-byte loads: a thread owns Vec<T>::N adjacent columns (n % N == 0 keeps every row 16-byte aligned) and
// keeps 8 rows in flight; each column still accumulates its rows in ascending order, so the result is bit-identical to
// k_vecmat_dense.  One step streams the whole matrix once: this is the HBM-bound kernel of run_markov.
template <typename T>
__global__ __launch_bounds__(256) void k_vecmat_dense_vec(const T *__restrict__ Tm, const double *__restrict__ x, double *__restrict__ part, int n)
{
    constexpr int N = Vec<T>::N;
    typedef T V __attribute__((ext_vector_type(N)));
    const int j = (blockIdx.x * blockDim.x + threadIdx.x) * N;
    if (j >= n) return;
    const int per = (n + gridDim.y - 1) / gridDim.y;
    const int i0 = blockIdx.y * per, i1 = min(n, i0 + per);
    double acc[N];
#pragma unroll
    for (int c = 0; c < N; ++c) acc[c] = 0.0;
    const T *p = Tm + (int64_t)i0 * n + j;
    int i = i0;
    for (; i + 8 <= i1; i += 8, p += (int64_t)8 * n) {
        V v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load((const V *)(p + (int64_t)u * n));
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double xi = x[i + u];
#pragma unroll
            for (int c = 0; c < N; ++c) acc[c] = fma(xi, (double)v[u][c], acc[c]);
        }
    }
    for (; i < i1; ++i, p += n) {
        const V v = *(const V *)p;
        const double xi = x[i];
#pragma unroll
        for (int c = 0; c < N; ++c) acc[c] = fma(xi, (double)v[c], acc[c]);
    }
#pragma unroll
    for (int c = 0; c < N; ++c) part[(int64_t)blockIdx.y * n + j + c] = acc[c];
}
__global__ void k_vecmat_reduce(const double *__restrict__ part, double *__restrict__ y, double *__restrict__ accum, int n, int nparts)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    double s = 0.0;
    for (int p = 0; p < nparts; ++p) s += part[(int64_t)p * n + j];
    y[j] = s;
    if (accum) accum[j] += s;     // path_integral: result = result + x   (diffusion.py:99)
}
// CSC form (column gather): y[j] = sum_p val[p] x[row[p]], one wave per column
template <typename T>
__global__ __launch_bounds__(256) void k_vecmat_csc(const int64_t *__restrict__ colptr, const int32_t *__restrict__ rowidx, const T *__restrict__ val,
                                                     const double *__restrict__ x, double *__restrict__ y, double *__restrict__ accum, int n)
{
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (j >= n) return;
    double acc = 0.0;
    for (int64_t p = colptr[j] + lane; p < colptr[j + 1]; p += 64) acc = fma((double)val[p], x[rowidx[p]], acc);
    acc = wave_sum(acc);
    if (lane == 0) { y[j] = acc; if (accum) accum[j] += acc; }
}

// ---------------------------------------------------------------- fit_gammas weight variants
// The non-default W constructions of VelocytoLoom.fit_gammas (analysis.py:1182-1192, 1208-1219),
// written densely so that vcy_fit_weighted(weight_mode 0) can consume them:
//   mode 0 "sum"            W = S/pa + U/pb                              (pa, pb = 99th percentiles)
//   mode 1 "prod"           W = (S/pa) * (U/pb)
//   mode 2 "maxmin_weighted" R = (clip(S, pa, pb) - pa) / (pb - pa);  W = 0.5 (R^pw + (1-R)^pw)
//   mode 3 "maxmin_double"  W = [Z<=pa | Z>=pb] + [S<=pc | S>=pd],  Z = S/sa + U/sb
template <typename T>
__global__ __launch_bounds__(256) void k_gamma_weights(const T *__restrict__ S, const T *__restrict__ U, T *__restrict__ W,
                                                        const double *__restrict__ pa, const double *__restrict__ pb,
                                                        const double *__restrict__ pc, const double *__restrict__ pd,
                                                        const double *__restrict__ sa, const double *__restrict__ sb, int64_t C, int G,
                                                        int64_t ld, int mode, double pw)
{
    const int64_t total = C * ld;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(t % ld);
        T w = T(0);
        if (g < G) {
            const double s = (double)S[t];
            if (mode == 0) w = (T)(s / pa[g] + (double)U[t] / pb[g]);
            else if (mode == 1) w = (T)((s / pa[g]) * ((double)U[t] / pb[g]));
            else if (mode == 2) {
                const double lo = pa[g], hi = pb[g];
                const double r = (fmin(fmax(s, lo), hi) - lo) / (hi - lo);
                w = (T)(0.5 * (pow(r, pw) + pow(1.0 - r, pw)));
            } else {
                const double z = (double)(T)(S[t] / (T)sa[g] + U[t] / (T)sb[g]);   // same T arithmetic as k_build_z
                w = (T)(((z <= pa[g] || z >= pb[g]) ? 1.0 : 0.0) + ((s <= pc[g] || s >= pd[g]) ? 1.0 : 0.0));
            }
        }
        W[t] = w;
    }
}

// ---------------------------------------------------------------- prepare_markov (analysis.py:1818-1863)
// One workgroup per row c of the dense (n, n) Markov matrix.  P arrives as CSR (the transition
// probabilities, or their transpose for direction="backwards"); emb is (n, edim) fp64.
//   t[c,j]  = P[c,j] * K(dist(c,j); sigma_D);  t[c,c] = max_j t[c,j];  t /= sum_j t      (:1853-1857)
//   kw[c,j] = K(dist(c,j); sigma_W) / sum_j K                                         (:1859-1860)
//   tr      = 0.8 t + 0.2 kw, rows renormalised                                        (:1861-1862)
// K(x; s) = exp(-x^2 / (2 s^2)) / sqrt(2 pi s^2)   (gaussian_kernel, :2449-2451)
__device__ __forceinline__ double gauss_k(double x, double s) { return exp(-(x * x) / (2.0 * s * s)) / sqrt(2.0 * 3.14159265358979323846 * s * s); }

template <typename T>
__global__ __launch_bounds__(256) void k_prepare_markov(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                         const double *__restrict__ pval, const double *__restrict__ emb, int edim,
                                                         T *__restrict__ tr, int n, double sigma_D, double sigma_W)
{
    __shared__ double red[8];
    __shared__ double s_vals[4];
    const int c = blockIdx.x, tid = threadIdx.x;
    double ec[4];
    for (int a = 0; a < edim; ++a) ec[a] = emb[(int64_t)c * edim + a];
    auto dist_to = [&](int j) {
        double d2 = 0.0;
        for (int a = 0; a < edim; ++a) { const double df = emb[(int64_t)j * edim + a] - ec[a]; d2 += df * df; }
        return sqrt(d2);
    };
    // dense noise kernel row sum
    double kw = 0.0;
    for (int j = tid; j < n; j += 256) kw += gauss_k(dist_to(j), sigma_W);
    kw = block_sum(kw, red);
    // sparse part: max and sum of P * K_D over the stored entries (zeros elsewhere -> max >= 0)
    const int64_t p0 = indptr[c], p1 = indptr[c + 1];
    double mx = 0.0, sm = 0.0;
    for (int64_t p = p0 + tid; p < p1; p += 256) {
        const int j = indices[p];
        const double v = pval[p] * gauss_k(dist_to(j), sigma_D);
        mx = fmax(mx, v);
        if (j != c) sm += v;
    }
    mx = wave_max(mx);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    sm = block_sum(sm, red) + mx;      // diagonal := row maximum
    T *row = tr + (int64_t)c * n;
    for (int j = tid; j < n; j += 256) row[j] = (T)(0.2 * gauss_k(dist_to(j), sigma_W) / kw);
    __syncthreads();
    for (int64_t p = p0 + tid; p < p1; p += 256) {
        const int j = indices[p];
        if (j != c) row[j] = (T)((double)row[j] + 0.8 * (pval[p] * gauss_k(dist_to(j), sigma_D)) / sm);
    }
    if (tid == 0) row[c] = (T)((double)row[c] + 0.8 * mx / sm);
    __syncthreads();
    double tot = 0.0;
    for (int j = tid; j < n; j += 256) tot += (double)row[j];
    tot = block_sum(tot, red);
    for (int j = tid; j < n; j += 256) row[j] = (T)((double)row[j] / tot);
    (void)s_vals;
}

// ---------------------------------------------------------------- the same chain in factored form
// tr[c,j] = (0.2 K_W(c,j) / kw[c] + s[c,j]) / tot[c]: a Gaussian of the embedding distance plus a sparse matrix s with the
// pattern of P and a diagonal.  Nothing in it needs n^2 memory: a step x <- x . tr is
//     y[j] = sum_c (x[c] / tot[c]) s[c,j]  +  sum_c u[c] exp2(-|es[c] - es[j]|^2),     u[c] = 0.2 g x[c] / (tot[c] kw[c]),
// es = embedding * sqrt(log2(e) / (2 sigma_W^2)), g = 1 / sqrt(2 pi sigma_W^2): a sparse product (k_vecmat_csc) and a discrete
// Gauss transform evaluated on the fly - n^2 exp2 per step (VALU-bound, 0.5 ms at 50 000 cells in f32) instead of n^2 matrix
// elements from HBM (1.6 ms from f32, 3.0 ms from f64 storage), and no 10-20 GB matrix.
// k_prepare_markov_factored: one workgroup per row c, same reductions as k_prepare_markov; sval in the CSR order of P
// (0 where P stores the diagonal), sdiag = 0.8 max / sm, kw, tot.
__global__ __launch_bounds__(256) void k_prepare_markov_factored(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                                  const double *__restrict__ pval, const double *__restrict__ emb, int edim,
                                                                  double *__restrict__ sval, double *__restrict__ sdiag, double *__restrict__ kw_out,
                                                                  double *__restrict__ tot_out, int n, double sigma_D, double sigma_W)
{
    __shared__ double red[8];
    const int c = blockIdx.x, tid = threadIdx.x;
    double ec[4];
    for (int a = 0; a < edim; ++a) ec[a] = emb[(int64_t)c * edim + a];
    auto dist_to = [&](int j) {
        double d2 = 0.0;
        for (int a = 0; a < edim; ++a) { const double df = emb[(int64_t)j * edim + a] - ec[a]; d2 += df * df; }
        return sqrt(d2);
    };
    double kw = 0.0;
    for (int j = tid; j < n; j += 256) kw += gauss_k(dist_to(j), sigma_W);
    kw = block_sum(kw, red);
    const int64_t p0 = indptr[c], p1 = indptr[c + 1];
    double mx = 0.0, sm = 0.0;
    for (int64_t p = p0 + tid; p < p1; p += 256) {
        const int j = indices[p];
        const double v = pval[p] * gauss_k(dist_to(j), sigma_D);
        mx = fmax(mx, v);
        if (j != c) sm += v;
    }
    mx = wave_max(mx);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    sm = block_sum(sm, red) + mx;      // diagonal := row maximum
    double tot = 0.0;
    for (int j = tid; j < n; j += 256) tot += 0.2 * gauss_k(dist_to(j), sigma_W) / kw;
    for (int64_t p = p0 + tid; p < p1; p += 256) {
        const int j = indices[p];
        const double v = j != c ? 0.8 * (pval[p] * gauss_k(dist_to(j), sigma_D)) / sm : 0.0;
        sval[p] = v;
        tot += v;
    }
    tot = block_sum(tot, red) + 0.8 * mx / sm;
    if (tid == 0) { sdiag[c] = 0.8 * mx / sm; kw_out[c] = kw; tot_out[c] = tot; }
}

template <typename CT>
__global__ void k_markov_scale_coords(const double *__restrict__ emb, CT *__restrict__ es, int64_t total, double scale)
{
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t < total) es[t] = (CT)(emb[t] * scale);
}
template <typename CT>
__global__ void k_markov_scale_x(const double *__restrict__ x, const double *__restrict__ tot, const double *__restrict__ kw, double *__restrict__ v,
                                 CT *__restrict__ u, int n, double coef, const int32_t *__restrict__ rank)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    const double vc = x[c] / tot[c];
    v[c] = vc;
    u[rank ? rank[c] : c] = (CT)(coef * vc / kw[c]);          // rank: the cell's position in the spatially sorted order of the culled transform
}

__device__ __forceinline__ float exp2_neg(float d2) { return __builtin_amdgcn_exp2f(-d2); }
// f64: 2^(-d2) for d2 >= 0 without the library's general exp2 (range checks, denormal paths: ~45 instructions).
// Round 4-5 form (tools/experiments/r06_exp2_poly13.patch keeps it for A/B): -d2 = k + r with k = rint(-d2), 2^r by the degree-13 Taylor
// polynomial of exp(r ln 2) on |r| <= 1/2, scaled by v_ldexp_f64 - 17 f64 instructions, 17 of the 22 of a (target, source) pair.
// Round 6 form (exp2_neg_tab): -d2 = (64 i + j) / 64 + r with |r| <= 2^-7,
//     2^(-d2) = 2^i  x  T[j]  x  2^r,        T[j] = 2^(j / 64) correctly rounded (64 doubles in LDS: entry j sits in bank pair j, so the 64
//                                            lanes of a read hit different banks or the same word - never a conflict),
//     2^r - 1 = r (c1 + r (c2 + r (c3 + r (c4 + r c5))))    Taylor of exp(r ln 2), truncation 3.5e-17 at |r| = 2^-7,
// the integer 64 i + j read out of the low word of fma(x, 64, 1.5 x 2^52) (no v_rndne, no conversion), 2^i applied by v_ldexp_f64 (the argument is
// clamped to -1000: below 2^-1000 the reference's exp() has long underflowed any sum it could enter): fmax, fma, sub, fma, four fma + a
// multiply, fma, ldexp = 11 f64 instructions and three 32-bit ones (22 % off a step of the full transform, 2.00 -> 1.56 ms at 50 000 cells), within
// 1 ulp of exp2() (tests/test_gpu_ops.py::test_f64_exp2_element_accuracy).  A NaN distance (a NaN or infinite coordinate) is dropped by the
// clamp - and does not need to survive it: such a cell's own normalisation kw[c] is NaN (it sums the same distances), so u[c] is NaN and
// fma(u[c], finite, .) poisons every target exactly as the dense chain does.
__constant__ double c_exp2_tab[64] = {                               // 2^(j / 64), j = 0 .. 63, correctly rounded (50-digit decimal arithmetic)
    0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,
    0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,
    0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
    0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,
    0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,
    0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,
    0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,
    0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
    0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,
    0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,
    0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
    0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,
    0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,
    0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
    0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0,
};
// (the table sits in LDS: every kernel that evaluates exp2_neg_tab fills it - 64 threads - before its first barrier)
__device__ __forceinline__ double exp2_neg_tab(double d2, const double *__restrict__ T)
{
    constexpr double MAGIC = 6755399441055744.0;                    // 1.5 x 2^52: the low word of x 64 + MAGIC is rint(64 x) in two's complement
    const double x = fmax(-d2, -1000.0);
    const double s = fma(x, 64.0, MAGIC);
    const int k = __double2loint(s);                                // 64 i + j
    const double r = fma(s - MAGIC, -0.015625, x);                  // exact, |r| <= 2^-7
    double q = 0.0013333558146428443;                               // ln2^5 / 5!
    q = fma(q, r, 0.009618129107628477);
    q = fma(q, r, 0.05550410866482158);
    q = fma(q, r, 0.2402265069591007);
    q = fma(q, r, 0.6931471805599453);
    q *= r;                                                         // 2^r - 1
    const double t = T[k & 63];
    const double w = fma(t, q, t);                                  // 2^(j / 64 + r) in [0.99, 2.02)
    return ldexp(w, k >> 6);                                        // (one v_ldexp_f64: the integer add on the exponent field cost two moves beside it)
}
// The same with a 256-entry table (2 KB: up to four lanes per bank pair) and a degree-4 polynomial on |r| <= 2^-9 (truncation 3.8e-17): one fma fewer.
// Measured (profiles/r06_markov_exp2.txt): the culled transform (one target per thread) gains 6 % - 0.610 -> 0.571 ms per step -, the full transform
// (two targets per thread: twice the table reads per wave-instruction of arithmetic) loses 5 % to bank conflicts: the culled kernel takes this form,
// the full one the 64-entry form.
__constant__ double c_exp2_tab256[256] = {
    0x1.0000000000000p+0, 0x1.00b1afa5abcbfp+0, 0x1.0163da9fb3335p+0, 0x1.02168143b0281p+0,
    0x1.02c9a3e778061p+0, 0x1.037d42e11bbccp+0, 0x1.04315e86e7f85p+0, 0x1.04e5f72f654b1p+0,
    0x1.059b0d3158574p+0, 0x1.0650a0e3c1f89p+0, 0x1.0706b29ddf6dep+0, 0x1.07bd42b72a836p+0,
    0x1.0874518759bc8p+0, 0x1.092bdf66607e0p+0, 0x1.09e3ecac6f383p+0, 0x1.0a9c79b1f3919p+0,
    0x1.0b5586cf9890fp+0, 0x1.0c0f145e46c85p+0, 0x1.0cc922b7247f7p+0, 0x1.0d83b23395decp+0,
    0x1.0e3ec32d3d1a2p+0, 0x1.0efa55fdfa9c5p+0, 0x1.0fb66affed31bp+0, 0x1.1073028d7233ep+0,
    0x1.11301d0125b51p+0, 0x1.11edbab5e2ab6p+0, 0x1.12abdc06c31ccp+0, 0x1.136a814f204abp+0,
    0x1.1429aaea92de0p+0, 0x1.14e95934f312ep+0, 0x1.15a98c8a58e51p+0, 0x1.166a45471c3c2p+0,
    0x1.172b83c7d517bp+0, 0x1.17ed48695bbc0p+0, 0x1.18af9388c8deap+0, 0x1.1972658375d2fp+0,
    0x1.1a35beb6fcb75p+0, 0x1.1af99f8138a1cp+0, 0x1.1bbe084045cd4p+0, 0x1.1c82f95281c6bp+0,
    0x1.1d4873168b9aap+0, 0x1.1e0e75eb44027p+0, 0x1.1ed5022fcd91dp+0, 0x1.1f9c18438ce4dp+0,
    0x1.2063b88628cd6p+0, 0x1.212be3578a819p+0, 0x1.21f49917ddc96p+0, 0x1.22bdda27912d1p+0,
    0x1.2387a6e756238p+0, 0x1.2451ffb82140ap+0, 0x1.251ce4fb2a63fp+0, 0x1.25e85711ece75p+0,
    0x1.26b4565e27cddp+0, 0x1.2780e341ddf29p+0, 0x1.284dfe1f56381p+0, 0x1.291ba7591bb70p+0,
    0x1.29e9df51fdee1p+0, 0x1.2ab8a66d10f13p+0, 0x1.2b87fd0dad990p+0, 0x1.2c57e39771b2fp+0,
    0x1.2d285a6e4030bp+0, 0x1.2df961f641589p+0, 0x1.2ecafa93e2f56p+0, 0x1.2f9d24abd886bp+0,
    0x1.306fe0a31b715p+0, 0x1.31432edeeb2fdp+0, 0x1.32170fc4cd831p+0, 0x1.32eb83ba8ea32p+0,
    0x1.33c08b26416ffp+0, 0x1.3496266e3fa2dp+0, 0x1.356c55f929ff1p+0, 0x1.36431a2de883bp+0,
    0x1.371a7373aa9cbp+0, 0x1.37f26231e754ap+0, 0x1.38cae6d05d866p+0, 0x1.39a401b7140efp+0,
    0x1.3a7db34e59ff7p+0, 0x1.3b57fbfec6cf4p+0, 0x1.3c32dc313a8e5p+0, 0x1.3d0e544ede173p+0,
    0x1.3dea64c123422p+0, 0x1.3ec70df1c5175p+0, 0x1.3fa4504ac801cp+0, 0x1.40822c367a024p+0,
    0x1.4160a21f72e2ap+0, 0x1.423fb2709468ap+0, 0x1.431f5d950a897p+0, 0x1.43ffa3f84b9d4p+0,
    0x1.44e086061892dp+0, 0x1.45c2042a7d232p+0, 0x1.46a41ed1d0057p+0, 0x1.4786d668b3237p+0,
    0x1.486a2b5c13cd0p+0, 0x1.494e1e192aed2p+0, 0x1.4a32af0d7d3dep+0, 0x1.4b17dea6db7d7p+0,
    0x1.4bfdad5362a27p+0, 0x1.4ce41b817c114p+0, 0x1.4dcb299fddd0dp+0, 0x1.4eb2d81d8abffp+0,
    0x1.4f9b2769d2ca7p+0, 0x1.508417f4531eep+0, 0x1.516daa2cf6642p+0, 0x1.5257de83f4eefp+0,
    0x1.5342b569d4f82p+0, 0x1.542e2f4f6ad27p+0, 0x1.551a4ca5d920fp+0, 0x1.56070dde910d2p+0,
    0x1.56f4736b527dap+0, 0x1.57e27dbe2c4cfp+0, 0x1.58d12d497c7fdp+0, 0x1.59c0827ff07ccp+0,
    0x1.5ab07dd485429p+0, 0x1.5ba11fba87a03p+0, 0x1.5c9268a5946b7p+0, 0x1.5d84590998b93p+0,
    0x1.5e76f15ad2148p+0, 0x1.5f6a320dceb71p+0, 0x1.605e1b976dc09p+0, 0x1.6152ae6cdf6f4p+0,
    0x1.6247eb03a5585p+0, 0x1.633dd1d1929fdp+0, 0x1.6434634ccc320p+0, 0x1.652b9febc8fb7p+0,
    0x1.6623882552225p+0, 0x1.671c1c70833f6p+0, 0x1.68155d44ca973p+0, 0x1.690f4b19e9538p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6b052fa75173ep+0, 0x1.6c012750bdabfp+0, 0x1.6cfdcddd47645p+0,
    0x1.6dfb23c651a2fp+0, 0x1.6ef9298593ae5p+0, 0x1.6ff7df9519484p+0, 0x1.70f7466f42e87p+0,
    0x1.71f75e8ec5f74p+0, 0x1.72f8286ead08ap+0, 0x1.73f9a48a58174p+0, 0x1.74fbd35d7cbfdp+0,
    0x1.75feb564267c9p+0, 0x1.77024b1ab6e09p+0, 0x1.780694fde5d3fp+0, 0x1.790b938ac1cf6p+0,
    0x1.7a11473eb0187p+0, 0x1.7b17b0976cfdbp+0, 0x1.7c1ed0130c132p+0, 0x1.7d26a62ff86f0p+0,
    0x1.7e2f336cf4e62p+0, 0x1.7f3878491c491p+0, 0x1.80427543e1a12p+0, 0x1.814d2add106d9p+0,
    0x1.82589994cce13p+0, 0x1.8364c1eb941f7p+0, 0x1.8471a4623c7adp+0, 0x1.857f4179f5b21p+0,
    0x1.868d99b4492edp+0, 0x1.879cad931a436p+0, 0x1.88ac7d98a6699p+0, 0x1.89bd0a478580fp+0,
    0x1.8ace5422aa0dbp+0, 0x1.8be05bad61778p+0, 0x1.8cf3216b5448cp+0, 0x1.8e06a5e0866d9p+0,
    0x1.8f1ae99157736p+0, 0x1.902fed0282c8ap+0, 0x1.9145b0b91ffc6p+0, 0x1.925c353aa2fe2p+0,
    0x1.93737b0cdc5e5p+0, 0x1.948b82b5f98e5p+0, 0x1.95a44cbc8520fp+0, 0x1.96bdd9a7670b3p+0,
    0x1.97d829fde4e50p+0, 0x1.98f33e47a22a2p+0, 0x1.9a0f170ca07bap+0, 0x1.9b2bb4d53fe0dp+0,
    0x1.9c49182a3f090p+0, 0x1.9d674194bb8d5p+0, 0x1.9e86319e32323p+0, 0x1.9fa5e8d07f29ep+0,
    0x1.a0c667b5de565p+0, 0x1.a1e7aed8eb8bbp+0, 0x1.a309bec4a2d33p+0, 0x1.a42c980460ad8p+0,
    0x1.a5503b23e255dp+0, 0x1.a674a8af46052p+0, 0x1.a799e1330b358p+0, 0x1.a8bfe53c12e59p+0,
    0x1.a9e6b5579fdbfp+0, 0x1.ab0e521356ebap+0, 0x1.ac36bbfd3f37ap+0, 0x1.ad5ff3a3c2774p+0,
    0x1.ae89f995ad3adp+0, 0x1.afb4ce622f2ffp+0, 0x1.b0e07298db666p+0, 0x1.b20ce6c9a8952p+0,
    0x1.b33a2b84f15fbp+0, 0x1.b468415b749b1p+0, 0x1.b59728de5593ap+0, 0x1.b6c6e29f1c52ap+0,
    0x1.b7f76f2fb5e47p+0, 0x1.b928cf22749e4p+0, 0x1.ba5b030a1064ap+0, 0x1.bb8e0b79a6f1fp+0,
    0x1.bcc1e904bc1d2p+0, 0x1.bdf69c3f3a207p+0, 0x1.bf2c25bd71e09p+0, 0x1.c06286141b33dp+0,
    0x1.c199bdd85529cp+0, 0x1.c2d1cd9fa652cp+0, 0x1.c40ab5fffd07ap+0, 0x1.c544778fafb22p+0,
    0x1.c67f12e57d14bp+0, 0x1.c7ba88988c933p+0, 0x1.c8f6d9406e7b5p+0, 0x1.ca3405751c4dbp+0,
    0x1.cb720dcef9069p+0, 0x1.ccb0f2e6d1675p+0, 0x1.cdf0b555dc3fap+0, 0x1.cf3155b5bab74p+0,
    0x1.d072d4a07897cp+0, 0x1.d1b532b08c968p+0, 0x1.d2f87080d89f2p+0, 0x1.d43c8eacaa1d6p+0,
    0x1.d5818dcfba487p+0, 0x1.d6c76e862e6d3p+0, 0x1.d80e316c98398p+0, 0x1.d955d71ff6075p+0,
    0x1.da9e603db3285p+0, 0x1.dbe7cd63a8315p+0, 0x1.dd321f301b460p+0, 0x1.de7d5641c0658p+0,
    0x1.dfc97337b9b5fp+0, 0x1.e11676b197d17p+0, 0x1.e264614f5a129p+0, 0x1.e3b333b16ee12p+0,
    0x1.e502ee78b3ff6p+0, 0x1.e653924676d76p+0, 0x1.e7a51fbc74c83p+0, 0x1.e8f7977cdb740p+0,
    0x1.ea4afa2a490dap+0, 0x1.eb9f4867cca6ep+0, 0x1.ecf482d8e67f1p+0, 0x1.ee4aaa2188510p+0,
    0x1.efa1bee615a27p+0, 0x1.f0f9c1cb6412ap+0, 0x1.f252b376bba97p+0, 0x1.f3ac948dd7274p+0,
    0x1.f50765b6e4540p+0, 0x1.f6632798844f8p+0, 0x1.f7bfdad9cbe14p+0, 0x1.f91d802243c89p+0,
    0x1.fa7c1819e90d8p+0, 0x1.fbdba3692d514p+0, 0x1.fd3c22b8f71f1p+0, 0x1.fe9d96b2a23d9p+0,
};
__device__ __forceinline__ double exp2_neg_tab256(double d2, const double *__restrict__ T)
{
    constexpr double MAGIC = 6755399441055744.0;
    const double x = fmax(-d2, -1000.0);
    const double s = fma(x, 256.0, MAGIC);
    const int k = __double2loint(s);
    const double r = fma(s - MAGIC, -0.00390625, x);                // |r| <= 2^-9
    double q = 0.009618129107628477;                                // ln2^4 / 4!
    q = fma(q, r, 0.05550410866482158);
    q = fma(q, r, 0.2402265069591007);
    q = fma(q, r, 0.6931471805599453);
    q *= r;
    const double t = T[k & 255];
    return ldexp(fma(t, q, t), k >> 8);
}
template <typename CT, int TAB = 64> __device__ __forceinline__ CT exp2_neg_any(CT d2, const double *T);
template <> __device__ __forceinline__ float exp2_neg_any<float, 64>(float d2, const double *) { return exp2_neg(d2); }
template <> __device__ __forceinline__ float exp2_neg_any<float, 256>(float d2, const double *) { return exp2_neg(d2); }
template <> __device__ __forceinline__ double exp2_neg_any<double, 64>(double d2, const double *T) { return exp2_neg_tab(d2, T); }
template <> __device__ __forceinline__ double exp2_neg_any<double, 256>(double d2, const double *T) { return exp2_neg_tab256(d2, T); }

// The sparse half of a step, y[j] = sum_p scsc[p] v[rowidx[p]] over column j (k_vecmat_csc), riding in the Gauss-transform launch: the two
// halves are independent, and a loop of thousands of steps is bound by the number of launches as soon as the kernels are small
// (13 us per kernel whatever it does at 10 000 cells).  The first `rows` rows of the grid (blockIdx.y) do it, four columns per workgroup.
struct SparseHalf {
    const int64_t *colptr;
    const int32_t *rowidx;
    const double *scsc, *v;
    double *y;
    int n, rows;                                                // rows = 0: no sparse half in this launch
};
__device__ __forceinline__ void sparse_half(const SparseHalf &sp)
{
    const int lane = threadIdx.x & 63;
    const int64_t j = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (j >= sp.n) return;
    double acc = 0.0;
    for (int64_t p = sp.colptr[j] + lane; p < sp.colptr[j + 1]; p += 64) acc = fma(sp.scsc[p], sp.v[sp.rowidx[p]], acc);
    acc = wave_sum(acc);
    if (lane == 0) sp.y[j] = acc;
}

// part[blockIdx.y][j] = sum over this block's source range of u[c] exp2(-|es[c] - es[j]|^2).  A thread owns JPT targets; the source
// index is uniform over the block (scalar loads).  8 terms are folded in CT before they are added to the fp64 accumulator.
template <typename CT, int EDIM, int JPT>
__global__ __launch_bounds__(256) void k_gauss_transform(const CT *__restrict__ es, const CT *__restrict__ u, double *__restrict__ part, int n,
                                                          int nparts, SparseHalf sp)
{
    __shared__ double s_tab[64];
    if ((int)blockIdx.y < sp.rows) { sparse_half(sp); return; }
    if (sizeof(CT) == 8) {                                      // the fp64 instance's table of 2^(j / 64) (exp2_neg_tab)
        if (threadIdx.x < 64) s_tab[threadIdx.x] = c_exp2_tab[threadIdx.x];
        __syncthreads();
    }
    const int by = (int)blockIdx.y - sp.rows;
    const int j0 = blockIdx.x * 256 * JPT + threadIdx.x;
    CT ej[JPT][EDIM];
#pragma unroll
    for (int t = 0; t < JPT; ++t) {
        const int j = min(j0 + t * 256, n - 1);
#pragma unroll
        for (int a = 0; a < EDIM; ++a) ej[t][a] = es[(int64_t)j * EDIM + a];
    }
    const int per = (n + nparts - 1) / nparts;
    const int c0 = by * per, c1 = min(n, c0 + per);
    double acc[JPT];
#pragma unroll
    for (int t = 0; t < JPT; ++t) acc[t] = 0.0;
    auto term = [&](int c, CT (&fold)[JPT]) {
        CT ec[EDIM];
#pragma unroll
        for (int a = 0; a < EDIM; ++a) ec[a] = es[(int64_t)c * EDIM + a];
        const CT uc = u[c];
#pragma unroll
        for (int t = 0; t < JPT; ++t) {
            CT d2 = CT(0);
#pragma unroll
            for (int a = 0; a < EDIM; ++a) { const CT df = ej[t][a] - ec[a]; d2 = fma(df, df, d2); }
            fold[t] = fma(uc, exp2_neg_any<CT>(d2, s_tab), fold[t]);
        }
    };
    int c = c0;
    for (; c + 8 <= c1; c += 8) {
        CT fold[JPT];
#pragma unroll
        for (int t = 0; t < JPT; ++t) fold[t] = CT(0);
#pragma unroll
        for (int q = 0; q < 8; ++q) term(c + q, fold);
#pragma unroll
        for (int t = 0; t < JPT; ++t) acc[t] += (double)fold[t];
    }
    {
        CT fold[JPT];
#pragma unroll
        for (int t = 0; t < JPT; ++t) fold[t] = CT(0);
        for (; c < c1; ++c) term(c, fold);
#pragma unroll
        for (int t = 0; t < JPT; ++t) acc[t] += (double)fold[t];
    }
#pragma unroll
    for (int t = 0; t < JPT; ++t)
        if (j0 + t * 256 < n) part[(int64_t)by * n + j0 + t * 256] = acc[t];
}
// The same transform for a kernel that is NARROW against the extent of the embedding (prepare_markov is typically called with sigma_W
// of a grid step): terms below 2^-cut of their weight are left out.  Cells are visited in a spatially sorted order (Hilbert curve of
// the embedding, made by the caller), so 256 consecutive targets and 32 consecutive sources both fill small boxes; a workgroup
// skips every chunk of 32 sources whose box is farther than sqrt(cut) from its targets' box.  The box tests of 64 chunks are made at
// once, one per lane, and the survivors walked through a ballot mask (one test after the other costs a scalar-load round trip each -
// more than the arithmetic they save).  What is left out per target is below n max(u) 2^-cut: cut = 48 (f32) / 72 (f64) puts it
// under the rounding of the accumulation itself.
constexpr int GT_CHUNK = 32;
template <typename CT, int EDIM>
__global__ void k_gauss_boxes(const CT *__restrict__ pts, int npts, CT *__restrict__ lo, CT *__restrict__ hi, int nbox)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;     // box b = bounding box of GT_CHUNK consecutive points
    if (b >= nbox) return;
    const int p0 = b * GT_CHUNK, p1 = min(npts, p0 + GT_CHUNK);
#pragma unroll
    for (int a = 0; a < EDIM; ++a) {
        CT l = pts[(int64_t)p0 * EDIM + a], h = l;
        for (int q = p0 + 1; q < p1; ++q) { const CT v = pts[(int64_t)q * EDIM + a]; l = fmin(l, v); h = fmax(h, v); }
        lo[(int64_t)b * EDIM + a] = l;
        hi[(int64_t)b * EDIM + a] = h;
    }
}

template <typename CT, int EDIM, int JPT>
__global__ __launch_bounds__(256) void k_gauss_transform_culled(const CT *__restrict__ es, const CT *__restrict__ u, double *__restrict__ part, int n,
                                                                 const CT *__restrict__ clo, const CT *__restrict__ chi, int nchunk, CT cut,
                                                                 int nparts, SparseHalf sp)
{
    __shared__ CT red[2][EDIM][4];
    __shared__ double s_tab[256];
    if ((int)blockIdx.y < sp.rows) { sparse_half(sp); return; }
    if (sizeof(CT) == 8) s_tab[threadIdx.x] = c_exp2_tab256[threadIdx.x];                       // 256 threads, published by the barrier below
    const int by = (int)blockIdx.y - sp.rows;
    const int j0 = blockIdx.x * 256 * JPT + threadIdx.x;
    CT ej[JPT][EDIM];
#pragma unroll
    for (int t = 0; t < JPT; ++t) {
        const int j = min(j0 + t * 256, n - 1);
#pragma unroll
        for (int a = 0; a < EDIM; ++a) ej[t][a] = es[(int64_t)j * EDIM + a];
    }
    CT tlo[EDIM], thi[EDIM];                                   // the box of this workgroup's targets
#pragma unroll
    for (int a = 0; a < EDIM; ++a) {
        CT l = ej[0][a], h = ej[0][a];
#pragma unroll
        for (int t = 1; t < JPT; ++t) { l = fmin(l, ej[t][a]); h = fmax(h, ej[t][a]); }
        l = -wave_max(-l);
        h = wave_max(h);
        if ((threadIdx.x & 63) == 0) { red[0][a][threadIdx.x >> 6] = l; red[1][a][threadIdx.x >> 6] = h; }
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < EDIM; ++a) {
        tlo[a] = fmin(fmin(red[0][a][0], red[0][a][1]), fmin(red[0][a][2], red[0][a][3]));
        thi[a] = fmax(fmax(red[1][a][0], red[1][a][1]), fmax(red[1][a][2], red[1][a][3]));
    }
    // the sources are split over blockIdx.y by chunks (finely: where the data is dense a few workgroups get all the work of a target
    // block, and the launch lasts as long as the busiest of them)
    const int qper = (nchunk + nparts - 1) / nparts;
    const int qa = by * qper, qb = min(nchunk, qa + qper);
    const int lane = threadIdx.x & 63;
    double acc[JPT];
#pragma unroll
    for (int t = 0; t < JPT; ++t) acc[t] = 0.0;
    auto term = [&](int c, CT (&fold)[JPT]) {
        CT ec[EDIM];
#pragma unroll
        for (int a = 0; a < EDIM; ++a) ec[a] = es[(int64_t)c * EDIM + a];
        const CT uc = u[c];
#pragma unroll
        for (int t = 0; t < JPT; ++t) {
            CT d2 = CT(0);
#pragma unroll
            for (int a = 0; a < EDIM; ++a) { const CT df = ej[t][a] - ec[a]; d2 = fma(df, df, d2); }
            fold[t] = fma(uc, exp2_neg_any<CT, 256>(d2, s_tab), fold[t]);
        }
    };
    for (int qq = qa; qq < qb; qq += 64) {
        const int mine = qq + lane;                            // every wave makes the same 64 tests and gets the same mask
        bool near = false;
        if (mine < qb) {
            CT d2 = CT(0);
#pragma unroll
            for (int a = 0; a < EDIM; ++a) {
                const CT gap = fmax(fmax(clo[(int64_t)mine * EDIM + a] - thi[a], tlo[a] - chi[(int64_t)mine * EDIM + a]), CT(0));
                d2 = fma(gap, gap, d2);
            }
            near = !(d2 > cut);
        }
        unsigned long long mask = __ballot(near);
        while (mask) {
            const int q = qq + __builtin_ctzll(mask);
            mask &= mask - 1;
            int c = q * GT_CHUNK;
            const int c1 = min(n, c + GT_CHUNK);
            for (; c + 8 <= c1; c += 8) {
                CT fold[JPT];
#pragma unroll
                for (int t = 0; t < JPT; ++t) fold[t] = CT(0);
#pragma unroll
                for (int k = 0; k < 8; ++k) term(c + k, fold);
#pragma unroll
                for (int t = 0; t < JPT; ++t) acc[t] += (double)fold[t];
            }
            if (c < c1) {
                CT fold[JPT];
#pragma unroll
                for (int t = 0; t < JPT; ++t) fold[t] = CT(0);
                for (; c < c1; ++c) term(c, fold);
#pragma unroll
                for (int t = 0; t < JPT; ++t) acc[t] += (double)fold[t];
            }
        }
    }
#pragma unroll
    for (int t = 0; t < JPT; ++t)
        if (j0 + t * 256 < n) part[(int64_t)by * n + j0 + t * 256] = acc[t];
}
// y[j] += the folded partials (fixed order); path_integral: accum += y.  Then, what the NEXT step starts with (k_markov_scale_x of
// the new y, saved a launch per step: thousands of steps of a few small kernels are bound by launches): v = y / tot, u = coef v / kw.
template <typename CT>
__global__ void k_gauss_reduce(const double *__restrict__ part, double *__restrict__ y, double *__restrict__ accum, int n, int nparts,
                               const int32_t *__restrict__ order, const double *__restrict__ tot, const double *__restrict__ kw,
                               double *__restrict__ v, CT *__restrict__ u, double coef)
{
    // order (culled transform): the partials are indexed by position in the sorted order; thread = position, so that the nparts reads
    // stay coalesced and only the update of y (and v) is scattered; u is indexed by position too
    const int jj = blockIdx.x * blockDim.x + threadIdx.x;
    if (jj >= n) return;
    const int j = order ? order[jj] : jj;
    double s = 0.0;
    for (int p = 0; p < nparts; ++p) s += part[(int64_t)p * n + jj];
    s += y[j];
    y[j] = s;
    if (accum) accum[j] += s;
    const double vj = s / tot[j];
    v[j] = vj;
    u[jj] = (CT)(coef * vj / kw[j]);
}

static inline int grid_for(int64_t total) { const int64_t b = (total + 255) / 256; return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); }
}  // namespace vcy

using namespace vcy;

extern "C" int vcy_row_sums(const void *M, double *out, int64_t C, int64_t G, int64_t ld, int dtype, vcy_stream stream)
{
    VCY_REQUIRE(M && out && C > 0 && G > 0 && ld >= G, "row_sums: bad arguments");
    const unsigned blocks = (unsigned)((C + 3) / 4);
    if (dtype == VCY_F32) hipLaunchKernelGGL(k_row_sums<float>, dim3(blocks), dim3(256), 0, as_stream(stream), (const float *)M, out, C, (int)G, ld);
    else if (dtype == VCY_F64) hipLaunchKernelGGL(k_row_sums<double>, dim3(blocks), dim3(256), 0, as_stream(stream), (const double *)M, out, C, (int)G, ld);
    else return fail(VCY_ERR_INVALID, "%s: bad dtype", "row_sums");
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}

extern "C" int vcy_scale_log(const void *M, const double *factor, void *out_sz, void *out_norm, int64_t C, int64_t G, int64_t ld,
                             double pcount, int fix_nonfinite, int dtype, vcy_stream stream)
{
    VCY_REQUIRE(M && (out_sz || out_norm) && C > 0 && G > 0 && ld >= G, "scale_log: bad arguments");
    const int blocks = grid_for(C * ld);
    if (dtype == VCY_F32) hipLaunchKernelGGL(k_scale_log<float>, dim3(blocks), dim3(256), 0, as_stream(stream), (const float *)M, factor, (float *)out_sz, (float *)out_norm, C, (int)G, ld, pcount, fix_nonfinite);
    else if (dtype == VCY_F64) hipLaunchKernelGGL(k_scale_log<double>, dim3(blocks), dim3(256), 0, as_stream(stream), (const double *)M, factor, (double *)out_sz, (double *)out_norm, C, (int)G, ld, pcount, fix_nonfinite);
    else return fail(VCY_ERR_INVALID, "%s: bad dtype", "scale_log");
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}

extern "C" int vcy_delta_transform(const void *hi_dim, const void *delta_S, void *dmat, void *e_out, int64_t C, int64_t G, int64_t ld,
                                   double used_dt, int mode, double psc, int dtype, vcy_stream stream)
{
    VCY_REQUIRE(hi_dim && delta_S && dmat && C > 0 && G > 0 && ld >= G, "delta_transform: bad arguments");
    VCY_REQUIRE(mode >= 0 && mode <= 3 && (mode != 3 || e_out), "delta_transform: bad mode (3 = logratio needs e_out)");
    const int blocks = grid_for(C * ld);
    if (dtype == VCY_F32) hipLaunchKernelGGL(k_delta_transform<float>, dim3(blocks), dim3(256), 0, as_stream(stream), (const float *)hi_dim, (const float *)delta_S, (float *)dmat, (float *)e_out, C, (int)G, ld, (float)used_dt, mode, (float)psc);
    else if (dtype == VCY_F64) hipLaunchKernelGGL(k_delta_transform<double>, dim3(blocks), dim3(256), 0, as_stream(stream), (const double *)hi_dim, (const double *)delta_S, (double *)dmat, (double *)e_out, C, (int)G, ld, used_dt, mode, psc);
    else return fail(VCY_ERR_INVALID, "%s: bad dtype", "delta_transform");
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}

extern "C" size_t vcy_permute_rows_nsign_workspace_bytes(int64_t C, int64_t G, int dtype)
{
    if (C <= 0 || G <= 0) return 0;
    return (size_t)C * (size_t)G * (dtype == VCY_F64 ? 8 : 4);
}

extern "C" int vcy_permute_rows_nsign(const void *in, void *out, void *workspace_a, void *workspace_b, int64_t C, int64_t G, int64_t ld,
                                      uint64_t seed, int dtype, vcy_stream stream)
{
    VCY_REQUIRE((workspace_a == nullptr) == (workspace_b == nullptr) && (!workspace_a || (workspace_a != workspace_b && workspace_a != in && workspace_b != in &&
                workspace_a != out && workspace_b != out)), "permute_rows_nsign: the two scratch buffers go together and are distinct from in / out");
    VCY_REQUIRE(in && out && in != out && C > 0 && G > 0 && ld >= G && C < (1ll << 30) && G < 65536ll * 256, "permute_rows_nsign: bad arguments");
    VCY_REQUIRE(dtype == VCY_F32 || dtype == VCY_F64, "permute_rows_nsign: bad dtype");
    int hb = 1;
    while ((1ll << (2 * hb)) < C) ++hb;                      // 4^hb >= C: at most 4 walks per cell on average, 1.3 at 50 000
    const uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
    hipStream_t st = as_stream(stream);
    if (workspace_a && G <= 65535) {                          // gene-major route: transpose, shuffle inside the rows, transpose back
        void *A = workspace_a, *B = workspace_b;
        int rc = vcy_transpose(in, A, C, G, ld, C, dtype, dtype, stream);
        if (rc) return rc;
        constexpr int CPT = 4;
        const dim3 grid((unsigned)((C + 256 * CPT - 1) / (256 * CPT)), (unsigned)G);
        if (dtype == VCY_F32) hipLaunchKernelGGL((k_permute_within_gene_rows<float, CPT>), grid, dim3(256), 0, st, (const float *)A, (float *)B, (int)C, lo, hi, hb);
        else hipLaunchKernelGGL((k_permute_within_gene_rows<double, CPT>), grid, dim3(256), 0, st, (const double *)A, (double *)B, (int)C, lo, hi, hb);
        VCY_LAUNCH_CHECK();
        return vcy_transpose(B, out, G, C, C, ld, dtype, dtype, stream);
    }
    constexpr int CPT = 8;
    const dim3 grid((unsigned)((ld + 255) / 256), (unsigned)((C + CPT - 1) / CPT));
    if (dtype == VCY_F32) hipLaunchKernelGGL((k_permute_rows_nsign<float, CPT>), grid, dim3(256), 0, st, (const float *)in, (float *)out, (int)C, (int)G, ld, lo, hi, hb);
    else hipLaunchKernelGGL((k_permute_rows_nsign<double, CPT>), grid, dim3(256), 0, st, (const double *)in, (double *)out, (int)C, (int)G, ld, lo, hi, hb);
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}

extern "C" int vcy_corr_fixup(void *vals, const int32_t *ixs, int64_t cell0, int64_t C_out, int64_t nrndm, int zero_self, int fix_nan,
                              double nan_to, int *nan_count, int dtype, vcy_stream stream)
{
    VCY_REQUIRE(vals && ixs && C_out > 0 && nrndm > 0, "corr_fixup: bad arguments");
    const int64_t total = C_out * nrndm;
    const int blocks = grid_for(total);
    if (dtype == VCY_F32) hipLaunchKernelGGL(k_corr_fixup<float>, dim3(blocks), dim3(256), 0, as_stream(stream), (float *)vals, ixs, cell0, total, (int)nrndm, zero_self, fix_nan, (float)nan_to, nan_count);
    else if (dtype == VCY_F64) hipLaunchKernelGGL(k_corr_fixup<double>, dim3(blocks), dim3(256), 0, as_stream(stream), (double *)vals, ixs, cell0, total, (int)nrndm, zero_self, fix_nan, nan_to, nan_count);
    else return fail(VCY_ERR_INVALID, "%s: bad dtype", "corr_fixup");
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}

extern "C" int vcy_transition_prob(const void *corr, const int32_t *ixs, const double *embedding, int edim, void *tp, void *wdiff,
                                   double *delta_embedding, int64_t cell0, int64_t C_out, int64_t n, double sigma_corr, int dtype,
                                   vcy_stream stream)
{
    VCY_REQUIRE(corr && ixs && embedding && delta_embedding && C_out > 0 && n > 0 && edim > 0 && edim <= 4 && sigma_corr > 0, "transition_prob: bad arguments");
    const unsigned blocks = (unsigned)((C_out + 3) / 4);
    if (dtype == VCY_F32) hipLaunchKernelGGL(k_transition_prob<float>, dim3(blocks), dim3(256), 0, as_stream(stream), (const float *)corr, ixs, embedding, edim, (float *)tp, (float *)wdiff, delta_embedding, cell0, C_out, (int)n, sigma_corr);
    else if (dtype == VCY_F64) hipLaunchKernelGGL(k_transition_prob<double>, dim3(blocks), dim3(256), 0, as_stream(stream), (const double *)corr, ixs, embedding, edim, (double *)tp, (double *)wdiff, delta_embedding, cell0, C_out, (int)n, sigma_corr);
    else return fail(VCY_ERR_INVALID, "%s: bad dtype", "transition_prob");
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}

extern "C" int vcy_row_cosproj(const void *A, const void *B, double *out, int64_t C, int64_t G, int64_t ld, int dtype, vcy_stream stream)
{
    VCY_REQUIRE(A && B && out && C > 0 && G > 0 && ld >= G, "row_cosproj: bad arguments");
    const unsigned blocks = (unsigned)((C + 3) / 4);
    if (dtype == VCY_F32) hipLaunchKernelGGL(k_row_cosproj<float>, dim3(blocks), dim3(256), 0, as_stream(stream), (const float *)A, (const float *)B, out, C, (int)G, ld);
    else if (dtype == VCY_F64) hipLaunchKernelGGL(k_row_cosproj<double>, dim3(blocks), dim3(256), 0, as_stream(stream), (const double *)A, (const double *)B, out, C, (int)G, ld);
    else return fail(VCY_ERR_INVALID, "%s: bad dtype", "row_cosproj");
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}

extern "C" size_t vcy_diffuse_workspace_bytes(int64_t n) { return (size_t)64 * (size_t)n * sizeof(double); }

extern "C" int vcy_diffuse_step_dense(const void *tr, const double *x, double *y, double *accum, void *workspace, int64_t n, int dtype,
                                      vcy_stream stream)
{
    VCY_REQUIRE(tr && x && y && workspace && n > 0 && x != y, "diffuse_step_dense: bad arguments");
    const int nparts = n >= 4096 ? 64 : (n >= 512 ? 16 : 1);
    dim3 grid((unsigned)((n + 255) / 256), nparts);
    hipStream_t st = as_stream(stream);
    VCY_REQUIRE(dtype == VCY_F32 || dtype == VCY_F64, "diffuse_step_dense: bad dtype");
    const int per_thread = dtype == VCY_F32 ? 4 : 2;
    if (n % per_thread == 0 && ((uintptr_t)tr & 15) == 0 && n >= 1024) {
        dim3 gv((unsigned)((n / per_thread + 255) / 256), nparts);
        if (dtype == VCY_F32) hipLaunchKernelGGL(k_vecmat_dense_vec<float>, gv, dim3(256), 0, st, (const float *)tr, x, (double *)workspace, (int)n);
        else hipLaunchKernelGGL(k_vecmat_dense_vec<double>, gv, dim3(256), 0, st, (const double *)tr, x, (double *)workspace, (int)n);
    }
    else if (dtype == VCY_F32) hipLaunchKernelGGL(k_vecmat_dense<float>, grid, dim3(256), 0, st, (const float *)tr, x, (double *)workspace, (int)n);
    else hipLaunchKernelGGL(k_vecmat_dense<double>, grid, dim3(256), 0, st, (const double *)tr, x, (double *)workspace, (int)n);
    VCY_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_vecmat_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const double *)workspace, y, accum, (int)n, nparts);
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}

extern "C" int vcy_diffuse_step_csc(const int64_t *colptr, const int32_t *rowidx, const void *val, const double *x, double *y,
                                    double *accum, int64_t n, int dtype, vcy_stream stream)
{
    VCY_REQUIRE(colptr && rowidx && val && x && y && n > 0 && x != y, "diffuse_step_csc: bad arguments");
    const unsigned blocks = (unsigned)((n + 3) / 4);
    if (dtype == VCY_F32) hipLaunchKernelGGL(k_vecmat_csc<float>, dim3(blocks), dim3(256), 0, as_stream(stream), colptr, rowidx, (const float *)val, x, y, accum, (int)n);
    else if (dtype == VCY_F64) hipLaunchKernelGGL(k_vecmat_csc<double>, dim3(blocks), dim3(256), 0, as_stream(stream), colptr, rowidx, (const double *)val, x, y, accum, (int)n);
    else return fail(VCY_ERR_INVALID, "%s: bad dtype", "diffuse_step_csc");
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}

extern "C" int vcy_gamma_weights(const void *S, const void *U, void *W, const double *pa, const double *pb, const double *pc,
                                 const double *pd, const double *sa, const double *sb, int64_t C, int64_t G, int64_t ld, int mode,
                                 double power, int dtype, vcy_stream stream)
{
    VCY_REQUIRE(S && W && pa && pb && C > 0 && G > 0 && ld >= G && mode >= 0 && mode <= 3, "gamma_weights: bad arguments");
    VCY_REQUIRE(mode == 2 || U, "gamma_weights: U needed");
    VCY_REQUIRE(mode != 3 || (pc && pd && sa && sb), "gamma_weights: maxmin_double needs pc, pd, sa, sb");
    const int blocks = grid_for(C * ld);
    if (dtype == VCY_F32) hipLaunchKernelGGL(k_gamma_weights<float>, dim3(blocks), dim3(256), 0, as_stream(stream), (const float *)S, (const float *)U, (float *)W, pa, pb, pc, pd, sa, sb, C, (int)G, ld, mode, power);
    else if (dtype == VCY_F64) hipLaunchKernelGGL(k_gamma_weights<double>, dim3(blocks), dim3(256), 0, as_stream(stream), (const double *)S, (const double *)U, (double *)W, pa, pb, pc, pd, sa, sb, C, (int)G, ld, mode, power);
    else return fail(VCY_ERR_INVALID, "%s: bad dtype", "gamma_weights");
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}

extern "C" int vcy_prepare_markov(const int64_t *indptr, const int32_t *indices, const double *pval, const double *embedding, int edim,
                                  void *tr, int64_t n, double sigma_D, double sigma_W, int dtype, vcy_stream stream)
{
    VCY_REQUIRE(indptr && indices && pval && embedding && tr && n > 0 && edim > 0 && edim <= 4 && sigma_D > 0 && sigma_W > 0, "prepare_markov: bad arguments");
    if (dtype == VCY_F32) hipLaunchKernelGGL(k_prepare_markov<float>, dim3((unsigned)n), dim3(256), 0, as_stream(stream), indptr, indices, pval, embedding, edim, (float *)tr, (int)n, sigma_D, sigma_W);
    else if (dtype == VCY_F64) hipLaunchKernelGGL(k_prepare_markov<double>, dim3((unsigned)n), dim3(256), 0, as_stream(stream), indptr, indices, pval, embedding, edim, (double *)tr, (int)n, sigma_D, sigma_W);
    else return fail(VCY_ERR_INVALID, "%s: bad dtype", "prepare_markov");
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}

// ---- factored chain (see k_prepare_markov_factored)
static inline int gauss_parts(int64_t n) { const int64_t bx = (n + 511) / 512; int p = (int)((3072 + bx - 1) / bx); return p < 1 ? 1 : (p > 64 ? 64 : p); }

extern "C" size_t vcy_markov_factored_workspace_bytes(int64_t n) { return (size_t)(64 + 2) * (size_t)(n > 0 ? n : 0) * sizeof(double); }

extern "C" int vcy_prepare_markov_factored(const int64_t *indptr, const int32_t *indices, const double *pval, const double *embedding, int edim,
                                           double *sval, double *sdiag, double *kw, double *tot, void *es, int64_t n, double sigma_D,
                                           double sigma_W, int compute_dtype, vcy_stream stream)
{
    VCY_REQUIRE(indptr && indices && pval && embedding && sval && sdiag && kw && tot && es, "prepare_markov_factored: null pointer");
    VCY_REQUIRE(n > 0 && n < (1ll << 31) && edim > 0 && edim <= 4 && sigma_D > 0 && sigma_W > 0, "prepare_markov_factored: bad arguments");
    VCY_REQUIRE(compute_dtype == VCY_F32 || compute_dtype == VCY_F64, "prepare_markov_factored: bad dtype");
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(k_prepare_markov_factored, dim3((unsigned)n), dim3(256), 0, st, indptr, indices, pval, embedding, edim, sval, sdiag, kw, tot,
                       (int)n, sigma_D, sigma_W);
    VCY_LAUNCH_CHECK();
    const double scale = sqrt(1.4426950408889634 / (2.0 * sigma_W * sigma_W));          // exp(-d^2 / (2 s^2)) = exp2(-(scale d)^2)
    const int64_t total = n * edim;
    if (compute_dtype == VCY_F32) hipLaunchKernelGGL(k_markov_scale_coords<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, embedding, (float *)es, total, scale);
    else hipLaunchKernelGGL(k_markov_scale_coords<double>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, embedding, (double *)es, total, scale);
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}

static inline int64_t gt_chunks(int64_t n) { return (n + GT_CHUNK - 1) / GT_CHUNK; }

template <typename CT>
static int diffuse_step_factored(const double *x, double *y, double *accum, const int64_t *colptr, const int32_t *rowidx, const double *scsc,
                                 const double *tot, const double *kw, const CT *es, int edim, double sigma_W, void *workspace, int64_t n,
                                 hipStream_t st, int prepared, const int32_t *rank = nullptr, const int32_t *order = nullptr,
                                 const CT *boxes = nullptr, double cut = 0.0)
{
    double *v = (double *)workspace;
    CT *u = (CT *)(v + n);
    double *part = v + 2 * n;
    const double coef = 0.2 / sqrt(2.0 * 3.14159265358979323846 * sigma_W * sigma_W);
    if (!prepared) {   // v = x / tot, u = coef v / kw; a step that follows another one on the same workspace finds them written by its fold
        hipLaunchKernelGGL(k_markov_scale_x<CT>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, tot, kw, v, u, (int)n, coef, rank);
        VCY_LAUNCH_CHECK();
    }
    // one launch for both halves of the step: the sparse product in the first rows of the grid, the Gauss transform in the others
    SparseHalf sp{colptr, rowidx, scsc, (const double *)v, y, (int)n, 0};
    const int64_t sparse_groups = (n + 3) / 4;
    int nparts = gauss_parts(n);
    if (boxes) {
        const int64_t nc = gt_chunks(n);
        nparts = nc < 64 ? (int)nc : 64;
        const CT *clo = boxes, *chi = clo + nc * edim;
        const unsigned gx = (unsigned)((n + 255) / 256);
        sp.rows = (int)((sparse_groups + gx - 1) / gx);
        dim3 gridc(gx, (unsigned)(sp.rows + nparts));
#define VCY_GTC(ED) hipLaunchKernelGGL((k_gauss_transform_culled<CT, ED, 1>), gridc, dim3(256), 0, st, es, (const CT *)u, part, (int)n, clo, chi, (int)nc, (CT)cut, nparts, sp)
        switch (edim) { case 1: VCY_GTC(1); break; case 2: VCY_GTC(2); break; case 3: VCY_GTC(3); break; default: VCY_GTC(4); break; }
#undef VCY_GTC
        VCY_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_gauss_reduce<CT>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const double *)part, y, accum, (int)n, nparts, order, tot, kw, v, u, coef);
        VCY_LAUNCH_CHECK();
        return VCY_OK;
    }
    const unsigned gx = (unsigned)((n + 511) / 512);
    sp.rows = (int)((sparse_groups + gx - 1) / gx);
    dim3 grid(gx, (unsigned)(sp.rows + nparts));
    switch (edim) {
    case 1: hipLaunchKernelGGL((k_gauss_transform<CT, 1, 2>), grid, dim3(256), 0, st, es, (const CT *)u, part, (int)n, nparts, sp); break;
    case 2: hipLaunchKernelGGL((k_gauss_transform<CT, 2, 2>), grid, dim3(256), 0, st, es, (const CT *)u, part, (int)n, nparts, sp); break;
    case 3: hipLaunchKernelGGL((k_gauss_transform<CT, 3, 2>), grid, dim3(256), 0, st, es, (const CT *)u, part, (int)n, nparts, sp); break;
    default: hipLaunchKernelGGL((k_gauss_transform<CT, 4, 2>), grid, dim3(256), 0, st, es, (const CT *)u, part, (int)n, nparts, sp); break;
    }
    VCY_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_gauss_reduce<CT>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const double *)part, y, accum, (int)n, nparts, order, tot, kw, v, u, coef);
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}

extern "C" size_t vcy_markov_cull_boxes_bytes(int64_t n, int edim, int compute_dtype)
{
    if (n <= 0 || edim <= 0) return 0;
    return (size_t)2 * (size_t)gt_chunks(n) * (size_t)edim * (compute_dtype == VCY_F64 ? 8 : 4);
}

template <typename CT>
static int markov_cull_boxes(const CT *es, CT *boxes, int64_t n, int edim, hipStream_t st)
{
    const int64_t nc = gt_chunks(n);
    CT *clo = boxes, *chi = clo + nc * edim;
#define VCY_GB(ED) hipLaunchKernelGGL((k_gauss_boxes<CT, ED>), dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, st, es, (int)n, clo, chi, (int)nc)
    switch (edim) { case 1: VCY_GB(1); break; case 2: VCY_GB(2); break; case 3: VCY_GB(3); break; default: VCY_GB(4); break; }
#undef VCY_GB
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}

extern "C" int vcy_markov_cull_boxes(const void *es_sorted, void *boxes, int64_t n, int edim, int compute_dtype, vcy_stream stream)
{
    VCY_REQUIRE(es_sorted && boxes && n > 0 && n < (1ll << 31) && edim > 0 && edim <= 4, "markov_cull_boxes: bad arguments");
    if (compute_dtype == VCY_F32) return markov_cull_boxes<float>((const float *)es_sorted, (float *)boxes, n, edim, as_stream(stream));
    if (compute_dtype == VCY_F64) return markov_cull_boxes<double>((const double *)es_sorted, (double *)boxes, n, edim, as_stream(stream));
    return fail(VCY_ERR_INVALID, "%s: bad dtype", "markov_cull_boxes");
}

extern "C" int vcy_diffuse_step_factored_culled(const double *x, double *y, double *accum, const int64_t *colptr, const int32_t *rowidx,
                                                const double *scsc, const double *tot, const double *kw, const void *es_sorted,
                                                const int32_t *rank, const int32_t *order, const void *boxes, int edim, double sigma_W,
                                                double cut, void *workspace, int64_t n, int prepared, int compute_dtype, vcy_stream stream)
{
    VCY_REQUIRE(x && y && colptr && rowidx && scsc && tot && kw && es_sorted && rank && order && boxes && workspace && x != y, "diffuse_step_factored_culled: bad arguments");
    VCY_REQUIRE(n > 0 && n < (1ll << 31) && edim > 0 && edim <= 4 && sigma_W > 0 && cut > 0, "diffuse_step_factored_culled: bad arguments");
    if (compute_dtype == VCY_F32) return diffuse_step_factored<float>(x, y, accum, colptr, rowidx, scsc, tot, kw, (const float *)es_sorted, edim, sigma_W, workspace, n, as_stream(stream), prepared, rank, order, (const float *)boxes, cut);
    if (compute_dtype == VCY_F64) return diffuse_step_factored<double>(x, y, accum, colptr, rowidx, scsc, tot, kw, (const double *)es_sorted, edim, sigma_W, workspace, n, as_stream(stream), prepared, rank, order, (const double *)boxes, cut);
    return fail(VCY_ERR_INVALID, "%s: bad dtype", "diffuse_step_factored_culled");
}

extern "C" int vcy_diffuse_step_factored(const double *x, double *y, double *accum, const int64_t *colptr, const int32_t *rowidx, const double *scsc,
                                         const double *tot, const double *kw, const void *es, int edim, double sigma_W, void *workspace, int64_t n,
                                         int prepared, int compute_dtype, vcy_stream stream)
{
    VCY_REQUIRE(x && y && colptr && rowidx && scsc && tot && kw && es && workspace && x != y, "diffuse_step_factored: bad arguments");
    VCY_REQUIRE(n > 0 && n < (1ll << 31) && edim > 0 && edim <= 4 && sigma_W > 0, "diffuse_step_factored: bad arguments");
    if (compute_dtype == VCY_F32) return diffuse_step_factored<float>(x, y, accum, colptr, rowidx, scsc, tot, kw, (const float *)es, edim, sigma_W, workspace, n, as_stream(stream), prepared);
    if (compute_dtype == VCY_F64) return diffuse_step_factored<double>(x, y, accum, colptr, rowidx, scsc, tot, kw, (const double *)es, edim, sigma_W, workspace, n, as_stream(stream), prepared);
    return fail(VCY_ERR_INVALID, "%s: bad dtype", "diffuse_step_factored");
}
