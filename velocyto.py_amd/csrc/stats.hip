// Per-gene statistics over cells for the pre-steps upstream of the hot path:
//   score_detection_levels  (analysis.py:456-475)  sum and number of expressing cells per gene
//   score_cv_vs_mean        (analysis.py:214-345)  mean / std(ddof=1), optionally of the winsorised (clipped) values
//   clusters_stats          (estimation.py:369-389) through a cell mask
// One streaming pass over the cells-major matrix (HBM-bound: C*G*s bytes read once): thread = one gene column,
// 256 adjacent genes per block (coalesced rows), grid.y = STATS_CB cell blocks whose fp64 partials a second kernel
// folds in fixed order (deterministic).
#include "common.h"

namespace vcy {

constexpr int STATS_CB = 64;
constexpr int STATS_NS = 4;       // sum, sum of squares, count(x > 0), max

template <typename T> __device__ __forceinline__ double load_as_double(const T *p) { return (double)*p; }

template <typename T>
__global__ __launch_bounds__(256) void k_gene_stats(const T *__restrict__ M, const double *__restrict__ cell_scale,
                                                     const double *__restrict__ lo, const double *__restrict__ hi,
                                                     const uint8_t *__restrict__ cell_mask, double *__restrict__ part,
                                                     int C, int G, int64_t ld)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    const int cb = blockIdx.y;
    const int per = (C + STATS_CB - 1) / STATS_CB;
    const int c0 = cb * per, c1 = min(C, c0 + per);
    const double l = lo ? lo[g] : -INFINITY, h = hi ? hi[g] : INFINITY;
    double s = 0.0, ss = 0.0, nz = 0.0, mx = -INFINITY;
    auto take = [&](double x, int c) {
        if (cell_mask && !cell_mask[c]) return;
        if (cell_scale) x *= cell_scale[c];
        x = fmin(fmax(x, l), h);               // np.clip(x, down, up)
        s += x;
        ss = fma(x, x, ss);
        nz += x > 0.0 ? 1.0 : 0.0;
        mx = fmax(mx, x);
    };
    int c = c0;
    for (; c + 3 < c1; c += 4) {
        double x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = load_as_double(M + (int64_t)(c + u) * ld + g);
#pragma unroll
        for (int u = 0; u < 4; ++u) take(x[u], c + u);
    }
    for (; c < c1; ++c) take(load_as_double(M + (int64_t)c * ld + g), c);
    double *p = part + (int64_t)cb * STATS_NS * G + g;
    p[0] = s; p[(int64_t)G] = ss; p[2 * (int64_t)G] = nz; p[3 * (int64_t)G] = mx;
}

__global__ void k_gene_stats_reduce(const double *__restrict__ part, double *__restrict__ stats, int G)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    double s = 0.0, ss = 0.0, nz = 0.0, mx = -INFINITY;
    for (int cb = 0; cb < STATS_CB; ++cb) {
        const double *p = part + (int64_t)cb * STATS_NS * G + g;
        s += p[0]; ss += p[(int64_t)G]; nz += p[2 * (int64_t)G]; mx = fmax(mx, p[3 * (int64_t)G]);
    }
    stats[g] = s; stats[(int64_t)G + g] = ss; stats[2 * (int64_t)G + g] = nz; stats[3 * (int64_t)G + g] = mx;
}

// Whole-matrix scale facts behind ops.partial_rules_for (the choice between VCY_RULES_PARTIAL and VCY_RULES_PARTIAL_NOPSC):
// sum |x| and the smallest non-zero |x| over the G logical columns of every row, one streaming pass (HBM-bound, C*G*s bytes).
// Non-zero is decided on the bit pattern (a denormal counts: v_rsq_f32 would read it as zero), the minimum is taken on the
// bits of |x| (monotone for non-negative floats).  Block partials -> one folding block, fixed order (deterministic).
constexpr int ABS_BLOCKS = 2048;

template <typename T>
__global__ __launch_bounds__(256) void k_abs_stats(const T *__restrict__ M, double *__restrict__ part, int C, int G, int64_t ld)
{
    using V = typename Vec<T>::type;
    constexpr int N = Vec<T>::N;
    __shared__ double red[3 * 4];
    const int nvec = (G + N - 1) / N;                       // rows are padded with zeros to ld >= G rounded up to 64 elements
    const int64_t total = (int64_t)C * nvec;
    double s = 0.0, mn = INFINITY, nz = 0.0;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t c = t / nvec;
        const int v = (int)(t - c * nvec);
        const V x = reinterpret_cast<const V *>(M + c * ld)[v];
        const T *xp = reinterpret_cast<const T *>(&x);
#pragma unroll
        for (int k = 0; k < N; ++k) {
            if (v * N + k >= G) continue;
            const double a = fabs((double)xp[k]);
            bool nonzero;
            if (sizeof(T) == 4) nonzero = (__float_as_uint((float)xp[k]) & 0x7fffffffu) != 0u;
            else nonzero = ((unsigned long long)__double_as_longlong((double)xp[k]) & 0x7fffffffffffffffull) != 0ull;
            s += a;
            if (nonzero) { mn = fmin(mn, a); nz += 1.0; }
        }
    }
    s = wave_sum(s); nz = wave_sum(nz);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mn = fmin(mn, __shfl_xor(mn, off, 64));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave] = s; red[4 + wave] = mn; red[8 + wave] = nz; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[3 * blockIdx.x] = red[0] + red[1] + red[2] + red[3];
        part[3 * blockIdx.x + 1] = fmin(fmin(red[4], red[5]), fmin(red[6], red[7]));
        part[3 * blockIdx.x + 2] = red[8] + red[9] + red[10] + red[11];
    }
}

__global__ __launch_bounds__(256) void k_abs_stats_fold(const double *__restrict__ part, double *__restrict__ out, int nblocks)
{
    __shared__ double red[3 * 4];
    double s = 0.0, mn = INFINITY, nz = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) { s += part[3 * b]; mn = fmin(mn, part[3 * b + 1]); nz += part[3 * b + 2]; }
    s = wave_sum(s); nz = wave_sum(nz);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mn = fmin(mn, __shfl_xor(mn, off, 64));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave] = s; red[4 + wave] = mn; red[8 + wave] = nz; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = red[0] + red[1] + red[2] + red[3];
        out[1] = fmin(fmin(red[4], red[5]), fmin(red[6], red[7]));
        out[2] = red[8] + red[9] + red[10] + red[11];
    }
}

}  // namespace vcy

using namespace vcy;

extern "C" int64_t vcy_gene_stats_workspace_bytes(int64_t G) { return (int64_t)STATS_CB * STATS_NS * G * (int64_t)sizeof(double); }

extern "C" int vcy_gene_stats(const void *M, const double *cell_scale, const double *lo, const double *hi, const uint8_t *cell_mask,
                              double *stats, void *workspace, int64_t C, int64_t G, int64_t ld, int dtype, vcy_stream stream)
{
    VCY_REQUIRE(M && stats && workspace, "gene_stats: null pointer");
    VCY_REQUIRE(C > 0 && G > 0 && ld >= G, "gene_stats: bad shape");
    VCY_REQUIRE((lo == nullptr) == (hi == nullptr), "gene_stats: lo and hi go together");
    hipStream_t st = as_stream(stream);
    dim3 grid((unsigned)((G + 255) / 256), STATS_CB);
    double *part = (double *)workspace;
    if (dtype == VCY_F32) hipLaunchKernelGGL(k_gene_stats<float>, grid, dim3(256), 0, st, (const float *)M, cell_scale, lo, hi, cell_mask, part, (int)C, (int)G, ld);
    else if (dtype == VCY_F64) hipLaunchKernelGGL(k_gene_stats<double>, grid, dim3(256), 0, st, (const double *)M, cell_scale, lo, hi, cell_mask, part, (int)C, (int)G, ld);
    else if (dtype == VCY_U16) hipLaunchKernelGGL(k_gene_stats<uint16_t>, grid, dim3(256), 0, st, (const uint16_t *)M, cell_scale, lo, hi, cell_mask, part, (int)C, (int)G, ld);
    else if (dtype == VCY_U8) hipLaunchKernelGGL(k_gene_stats<uint8_t>, grid, dim3(256), 0, st, (const uint8_t *)M, cell_scale, lo, hi, cell_mask, part, (int)C, (int)G, ld);
    else return fail(VCY_ERR_INVALID, "%s: bad dtype", "gene_stats");
    VCY_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_gene_stats_reduce, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, st, (const double *)part, stats, (int)G);
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}

extern "C" int64_t vcy_abs_stats_workspace_bytes(void) { return (int64_t)ABS_BLOCKS * 3 * (int64_t)sizeof(double); }

extern "C" int vcy_abs_stats(const void *M, double *out3, void *workspace, int64_t C, int64_t G, int64_t ld, int dtype, vcy_stream stream)
{
    VCY_REQUIRE(M && out3 && workspace, "abs_stats: null pointer");
    VCY_REQUIRE(C > 0 && G > 0 && ld >= G, "abs_stats: bad shape");
    VCY_REQUIRE(dtype == VCY_F32 || dtype == VCY_F64, "abs_stats: bad dtype");
    VCY_REQUIRE(ld % (dtype == VCY_F32 ? 4 : 2) == 0 && (uintptr_t)M % 16 == 0, "abs_stats: rows must be 16-byte aligned");
    VCY_REQUIRE(ld >= (G + (dtype == VCY_F32 ? 3 : 1)) / (dtype == VCY_F32 ? 4 : 2) * (dtype == VCY_F32 ? 4 : 2), "abs_stats: the last vector of a row must lie inside ld");
    hipStream_t st = as_stream(stream);
    const int64_t nvec = (G + (dtype == VCY_F32 ? 3 : 1)) / (dtype == VCY_F32 ? 4 : 2);
    int blocks = (int)((C * nvec + 255) / 256 > ABS_BLOCKS ? ABS_BLOCKS : (C * nvec + 255) / 256);
    double *part = (double *)workspace;
    if (dtype == VCY_F32) hipLaunchKernelGGL(k_abs_stats<float>, dim3(blocks), dim3(256), 0, st, (const float *)M, part, (int)C, (int)G, ld);
    else hipLaunchKernelGGL(k_abs_stats<double>, dim3(blocks), dim3(256), 0, st, (const double *)M, part, (int)C, (int)G, ld);
    VCY_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_abs_stats_fold, dim3(1), dim3(256), 0, st, (const double *)part, out3, blocks);
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}
