// pool.hip -- stage A: kNN pooling of the count matrices (the data @ w.T of
// neighbors.convolve_by_sparse_weights, neighbors.py:416-423; analysis.py:1011-1019).
//
// out[c,:] = sum_p w[p] * data[indices[p],:]: a gather-average of k+1 neighbour gene vectors.
// Every cell's vector is gathered by ~k other cells, so naive gathering moves (k+1) * G * s
// bytes per cell.  The launch therefore walks GENE SLABS: blocks are ordered slab-major
// (all cells of slab 0, then slab 1, ...) and a slab of all cells (C * slab * s bytes, ~100 MB
// at 50k cells x 512 genes) stays resident in the 256 MiB Infinity Cache while it is gathered
// k+1 times -- HBM then sees ~2 * G * s per cell (read once, write once), the algorithmic figure.
// The adjacency has k/C ~ 0.06 % density with scattered columns, so this is not a dense block
// contraction and MFMA is deliberately not used (it would multiply by zeros).
#include "common.h"

namespace vcy {

template <typename T>
__global__ __launch_bounds__(256) void k_knn_pool(const T *__restrict__ data, T *__restrict__ out, const int64_t *__restrict__ indptr,
                                                   const int32_t *__restrict__ indices, const T *__restrict__ w,
                                                   const int32_t *__restrict__ order, int G, int64_t ld,
                                                   int64_t cell0, int C_out, int slab, int maximum)
{
    using V = typename Vec<T>::type;
    constexpr int N = Vec<T>::N;
    // slab-major block order; inside a slab the schedule is XCD-aware: workgroup b lands on XCD b % 8
    // (observed dispatch rule, used for speed only), so XCD x is given the contiguous schedule range
    // [x*per, (x+1)*per): neighbouring cells - which share neighbours - hit the SAME per-XCD L2.
    const int per = (C_out + 7) / 8, nblk = per * 8;
    const int64_t b = blockIdx.x;
    const int s = (int)(b / nblk), bi = (int)(b % nblk);
    const int pos = (bi & 7) * per + (bi >> 3);
    if (pos >= C_out) return;
    const int cl = order ? order[pos] : pos;                 // schedule position -> cell
    const int g0 = s * slab, g1 = min(G, g0 + slab);
    const int64_t p0 = indptr[cl], p1 = indptr[cl + 1];
    const int nvec = (g1 - g0) / N;                        // slab and ld are multiples of N; tail handled below
    for (int v = threadIdx.x; v < nvec; v += blockDim.x) {
        T acc[N];
#pragma unroll
        for (int k = 0; k < N; ++k) acc[k] = T(0);
        int64_t p = p0;
        for (; p + 3 < p1; p += 4) {                       // 4 gathers in flight
            V x[4]; T ww[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                x[u] = reinterpret_cast<const V *>(data + (int64_t)indices[p + u] * ld + g0)[v];
                ww[u] = w[p + u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const T *xp = reinterpret_cast<const T *>(&x[u]);
#pragma unroll
                for (int k = 0; k < N; ++k) acc[k] = fma(ww[u], xp[k], acc[k]);
            }
        }
        for (; p < p1; ++p) {
            const V xv = reinterpret_cast<const V *>(data + (int64_t)indices[p] * ld + g0)[v];
            const T wv = w[p];
            const T *xp = reinterpret_cast<const T *>(&xv);
#pragma unroll
            for (int k = 0; k < N; ++k) acc[k] = fma(wv, xp[k], acc[k]);
        }
        if (maximum) {
            const V sv = reinterpret_cast<const V *>(data + (cell0 + cl) * ld + g0)[v];
            const T *sp = reinterpret_cast<const T *>(&sv);
#pragma unroll
            for (int k = 0; k < N; ++k) acc[k] = acc[k] > sp[k] ? acc[k] : sp[k];
        }
        V o;
        T *op = reinterpret_cast<T *>(&o);
#pragma unroll
        for (int k = 0; k < N; ++k) op[k] = acc[k];
        reinterpret_cast<V *>(out + (int64_t)cl * ld + g0)[v] = o;
    }
    for (int g = g0 + nvec * N + threadIdx.x; g < g1; g += blockDim.x) {   // < N tail genes of the last slab
        T a = T(0);
        for (int64_t p = p0; p < p1; ++p) a = fma(w[p], data[(int64_t)indices[p] * ld + g], a);
        if (maximum) { const T sv = data[(cell0 + cl) * ld + g]; a = a > sv ? a : sv; }
        out[(int64_t)cl * ld + g] = a;
    }
}
}  // namespace vcy

using namespace vcy;

extern "C" int vcy_knn_pool(const void *data, void *out, const int64_t *indptr, const int32_t *indices, const void *w,
                            const int32_t *order, int64_t C, int64_t G, int64_t ld, int64_t cell0, int64_t C_out, int maximum,
                            int64_t slab_genes, int dtype, vcy_stream stream)
{
    VCY_REQUIRE(data && out && indptr && indices && w, "knn_pool: null pointer");
    VCY_REQUIRE(C > 0 && G > 0 && ld >= G && C_out > 0 && cell0 >= 0 && cell0 + C_out <= C, "knn_pool: bad shape");
    VCY_REQUIRE(dtype == VCY_F32 || dtype == VCY_F64, "knn_pool: bad dtype");
    const int N = dtype == VCY_F32 ? 4 : 2;
    VCY_REQUIRE(ld % N == 0, "knn_pool: ld must keep rows 16-byte aligned");
    VCY_REQUIRE(data != out, "knn_pool: in-place pooling is not supported");
    int64_t slab = slab_genes > 0 ? slab_genes : 512;
    slab = (slab + N - 1) / N * N;
    if (slab > G) slab = (G + N - 1) / N * N;
    const int64_t nslab = (G + slab - 1) / slab;
    const int threads = slab / N >= 256 ? 256 : (slab / N >= 128 ? 128 : 64);
    const int64_t blocks = nslab * ((C_out + 7) / 8 * 8);
    VCY_REQUIRE(blocks < (1LL << 31), "knn_pool: grid too large");
    hipStream_t st = as_stream(stream);
    if (dtype == VCY_F32)
        hipLaunchKernelGGL(k_knn_pool<float>, dim3((unsigned)blocks), dim3(threads), 0, st, (const float *)data, (float *)out, indptr,
                           indices, (const float *)w, order, (int)G, ld, cell0, (int)C_out, (int)slab, maximum);
    else
        hipLaunchKernelGGL(k_knn_pool<double>, dim3((unsigned)blocks), dim3(threads), 0, st, (const double *)data, (double *)out, indptr,
                           indices, (const double *)w, order, (int)G, ld, cell0, (int)C_out, (int)slab, maximum);
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}
