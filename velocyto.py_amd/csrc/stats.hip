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
