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
// VALU at ~1/3 of its rate when HBM runs at 6 TB/s, so the kernel is HBM-bound by design.
#include "common.h"

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
    T r;
    if (TR == VCY_LOG10 && RULES == VCY_RULES_PARTIAL) r = (t >= T(0)) ? s : -s;
    else r = (t > T(0)) ? s : -s;
    if (TR == VCY_SQRT && RULES == VCY_RULES_PARTIAL) r = (fabs(t) < T(1e-16)) ? T(0) : r;
    return r;
}

template <typename T> __device__ __forceinline__ T pearson_from_moments(double sA, double sAA, double sAb, double sb, double sbb, double n)
{
    const double cov = sAb - sA * sb / n;
    const double va = sAA - sA * sA / n;
    const double vb = sbb - sb * sb / n;
    return (T)(cov / sqrt(va * vb));  // va == 0 -> 0/0 = NaN, like the reference's 0 * inf
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
                                                       int64_t cell0, int64_t d_row0, int nrndm, int gchunk, T psc)
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
    const int cl = order ? order[blockIdx.x] : (int)blockIdx.x;  // local output row
    const int64_t c = cell0 + cl;
    const T *erow_c = e + c * ld;
    const T *drow_c = d + (c - d_row0) * ld;

    for (int n = tid; n < 3 * nrndm; n += blockDim.x) acc[n] = T(0);
    double sb = 0.0, sbb = 0.0;

    for (int g0 = 0; g0 < G; g0 += gchunk) {
        const int gl = min(gchunk, G - g0);
        const int nvec = gl / N;
        __syncthreads();  // previous chunk fully consumed (also orders the acc zeroing)
        for (int v = tid; v < nvec; v += blockDim.x) {
            const V ev = reinterpret_cast<const V *>(erow_c + g0)[v];
            const V dv = reinterpret_cast<const V *>(drow_c + g0)[v];
            reinterpret_cast<V *>(ec)[v] = ev;
            reinterpret_cast<V *>(dc)[v] = dv;
            const T *dp = reinterpret_cast<const T *>(&dv);
#pragma unroll
            for (int k = 0; k < N; ++k) { sb += (double)dp[k]; sbb += (double)dp[k] * (double)dp[k]; }
        }
        for (int g = nvec * N + tid; g < gl; g += blockDim.x) {  // < N scalar tail elements
            const T ev = erow_c[g0 + g], dv = drow_c[g0 + g];
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
                        const T a = xform<T, TR, RULES>(xp[k] - ep[k], psc);
                        sA[k] += a;
                        sAA[k] = fma(a, a, sAA[k]);
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
                    const T a = xform<T, TR, RULES>(xp[k] - ep[k], psc);
                    sA[k] += a;
                    sAA[k] = fma(a, a, sAA[k]);
                    sAb[k] = fma(a, bp[k], sAb[k]);
                }
            }
            {
                const int g = nvec * N + lane;
                if (g < gl) {
                    const T a = xform<T, TR, RULES>(row[g] - ec[g], psc);
                    sA[0] += a;
                    sAA[0] = fma(a, a, sAA[0]);
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
                const T a = xform<T, TR, VCY_RULES_FULL>(x - ecs[cg * 4 + k][g], psc);
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
static int g_lds_budget = 0;   // usable dynamic LDS per workgroup
static int g_cus = 0;

static int query_device()
{
    if (g_cus) return VCY_OK;
    int dev = 0;
    VCY_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    VCY_CHECK_HIP(hipGetDeviceProperties(&p, dev));
    g_cus = p.multiProcessorCount;
    g_lds_budget = (int)p.sharedMemPerBlock;   // 64 KiB default; opt-in up to 160 KiB on gfx950
    int maxopt = 0;
    if (hipDeviceGetAttribute(&maxopt, hipDeviceAttributeSharedMemPerBlockOptin, dev) == hipSuccess && maxopt > g_lds_budget)
        g_lds_budget = maxopt;
    return VCY_OK;
}

template <typename T, int TR, int RULES>
static int launch_partial(const void *e, const void *d, const int32_t *ixs, void *out, const int32_t *order, int64_t G,
                          int64_t ld, int64_t cell0, int64_t C_out, int64_t d_row0, int64_t nrndm, double psc, hipStream_t st)
{
    constexpr int N = Vec<T>::N;
    const int quantum = 64 * N;  // one wave-instruction worth of elements
    const size_t fixed = sizeof(T) * 3 * ((nrndm + 1) & ~1) + 32 * sizeof(double);
    // budget: stay under 150 KiB so one workgroup (16 waves) owns a CU; fewest chunks that fit
    const size_t budget = (size_t)(g_lds_budget > 153600 ? 153600 : g_lds_budget);
    if (fixed + 2 * quantum * sizeof(T) > budget)
        return fail(VCY_ERR_UNSUPPORTED, "%s: nrndm=%lld needs more LDS than the %lld-byte budget", "coldeltacor_partial", (long long)nrndm, (long long)budget);
    const int64_t max_chunk = (int64_t)((budget - fixed) / (2 * sizeof(T))) / quantum * quantum;
    const int64_t Gq = (G + quantum - 1) / quantum * quantum;
    const int64_t nchunks = (Gq + max_chunk - 1) / max_chunk;
    int64_t gchunk = ((Gq / quantum + nchunks - 1) / nchunks) * quantum;
    const size_t lds = fixed + 2 * gchunk * sizeof(T);
    auto kern = k_cdc_partial<T, TR, RULES>;
    VCY_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int threads = (nrndm >= 16) ? 1024 : (nrndm >= 8 ? 512 : 256);
    hipLaunchKernelGGL(kern, dim3((unsigned)C_out), dim3(threads), lds, st, (const T *)e, (const T *)d, ixs, (T *)out, order,
                       (int)G, ld, cell0, d_row0, (int)nrndm, (int)gchunk, (T)psc);
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}

template <typename T>
static int dispatch_partial(const void *e, const void *d, const int32_t *ixs, void *out, const int32_t *order, int64_t G,
                            int64_t ld, int64_t cell0, int64_t C_out, int64_t d_row0, int64_t nrndm, int transform, int rules,
                            double psc, hipStream_t st)
{
#define VCY_CASE(TR, RU) \
    if (transform == TR && rules == RU) return launch_partial<T, TR, RU>(e, d, ixs, out, order, G, ld, cell0, C_out, d_row0, nrndm, psc, st);
    VCY_CASE(VCY_LINEAR, VCY_RULES_PARTIAL)
    VCY_CASE(VCY_LINEAR, VCY_RULES_FULL)
    VCY_CASE(VCY_SQRT, VCY_RULES_PARTIAL)
    VCY_CASE(VCY_SQRT, VCY_RULES_FULL)
    VCY_CASE(VCY_LOG10, VCY_RULES_PARTIAL)
    VCY_CASE(VCY_LOG10, VCY_RULES_FULL)
#undef VCY_CASE
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

extern "C" int vcy_coldeltacor_partial(const void *e, const void *d, const int32_t *ixs, void *out, const int32_t *order,
                                       int64_t C, int64_t G, int64_t ld, int64_t cell0, int64_t C_out, int64_t d_row0,
                                       int64_t nrndm, int transform, int rules, double psc, int dtype, vcy_stream stream)
{
    VCY_REQUIRE(e && d && ixs && out, "coldeltacor_partial: null pointer");
    VCY_REQUIRE(C > 0 && G > 0 && nrndm > 0 && C_out > 0 && cell0 >= 0 && cell0 + C_out <= C, "coldeltacor_partial: bad shape");
    VCY_REQUIRE(ld >= G, "coldeltacor_partial: ld < G");
    VCY_REQUIRE(d_row0 >= 0 && d_row0 <= cell0, "coldeltacor_partial: d must cover cells cell0..cell0+C_out-1");
    VCY_REQUIRE(dtype == VCY_F32 || dtype == VCY_F64, "coldeltacor_partial: bad dtype");
    VCY_REQUIRE(ld % (dtype == VCY_F32 ? 4 : 2) == 0, "coldeltacor_partial: ld must keep rows 16-byte aligned");
    VCY_REQUIRE(((uintptr_t)e % 16 == 0) && ((uintptr_t)d % 16 == 0), "coldeltacor_partial: e/d must be 16-byte aligned");
    int rc = query_device();
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    if (dtype == VCY_F32) return dispatch_partial<float>(e, d, ixs, out, order, G, ld, cell0, C_out, d_row0, nrndm, transform, rules, psc, st);
    return dispatch_partial<double>(e, d, ixs, out, order, G, ld, cell0, C_out, d_row0, nrndm, transform, rules, psc, st);
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
