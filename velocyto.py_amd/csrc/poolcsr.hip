// poolcsr.hip -- stage A at atlas scale: kNN pooling straight from SPARSE (CSR) count layers.
//
// Reference: the same product as pool.hip, data @ w.T of neighbors.convolve_by_sparse_weights (neighbors.py:416-423)
// applied to S_sz = factor * S (analysis.py:546-549, 1011-1016), for datasets whose count layers do not fit as dense
// matrices (BASELINE.json configs[4]: 1M cells x 30k genes at ~8 % density; the reference itself loads every layer dense,
// analysis.py:59-61, which is what caps it).  A cell's row is (gene index int32, count uint8/uint16) pairs, genes ascending.
//
// out[c, :] = sum_p (w[p] * scale[j_p]) * counts[j_p, :]  over the k+1 graph entries p of cell c:  a MERGE of k+1 sparse
// rows into one dense row.  One WAVE owns a (cell, gene slab) unit: the slab (2048 genes, 8 KB of f32) lives in LDS, the
// graph entries are walked in order and every non-zero of the row's slab segment does  slab[g] = fma(ws, x, slab[g]).
// LDS operations of one wave execute in program order and the non-zeros of one row hit distinct genes, so every gene
// receives its contributions in graph order with one rounding each - bit for bit the arithmetic of the dense kernel
// (k_knn_pool_counts: the zeros it multiplies add +0), which is what lets the tests demand equality with the dense path.
// Slab segments of a row are found through a per-row table of slab boundaries (vcy_csr_slab_ptr, built once per
// layer).  Bytes: a non-zero costs 5-6 B (index + count) instead of G bytes per row; the output is the same dense f32
// row (HBM write floor: G * 4 B per cell and layer).
#include "common.h"

namespace vcy {

constexpr int CSR_SLAB = 2048;        // genes per unit: 8 KB (f32) / 16 KB (f64) of LDS per wave
constexpr int CSR_WAVES = 4;          // waves (units in flight) per workgroup
constexpr int CSR_NZ = 4;             // non-zeros per lane and row in flight (4 x 64 = 256 per pass over a row segment)

// slabptr[r * (nslab + 1) + s] = number of non-zeros of row r with gene < s * slab  (s = 0 .. nslab): lower bounds
__global__ __launch_bounds__(256) void k_csr_slab_ptr(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                       int32_t *__restrict__ slabptr, int64_t C, int nslab, int slab)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= C * (nslab + 1)) return;
    const int64_t r = t / (nslab + 1);
    const int s = (int)(t - r * (nslab + 1));
    const int64_t rs = indptr[r];
    const int n = (int)(indptr[r + 1] - rs);
    const int32_t key = s * slab;
    int lo = 0, hi = n;                                  // first position with indices[rs + pos] >= key
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (indices[rs + mid] < key) lo = mid + 1; else hi = mid;
    }
    slabptr[t] = lo;
}

template <typename T, typename CT>
__global__ __launch_bounds__(64 * CSR_WAVES) void k_knn_pool_csr(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                                  const CT *__restrict__ data, const int32_t *__restrict__ slabptr,
                                                                  const double *__restrict__ scale, T *__restrict__ out,
                                                                  const int64_t *__restrict__ g_indptr, const int32_t *__restrict__ g_indices,
                                                                  const T *__restrict__ w, const int32_t *__restrict__ order, int G, int64_t ld_out,
                                                                  int64_t cell0, int C_out, int nslab, int maximum)
{
    using V = typename Vec<T>::type;
    constexpr int N = Vec<T>::N;
    __shared__ __attribute__((aligned(16))) T lds[CSR_WAVES][CSR_SLAB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T *slab = lds[wave];
    // slab-major, XCD-aware schedule as in pool.hip: workgroup b runs on XCD b % 8 (observed; speed only), so XCD x is given a
    // contiguous range of the (locality-sorted) cell schedule and co-resident units re-read the same CSR rows from one L2
    const int nquads = (C_out + CSR_WAVES - 1) / CSR_WAVES, per = (nquads + 7) / 8, nblk = per * 8;
    const int64_t b = blockIdx.x;
    const int s = (int)(b / nblk), bi = (int)(b % nblk);
    const int pos = ((bi & 7) * per + (bi >> 3)) * CSR_WAVES + wave;
    if ((bi & 7) * per + (bi >> 3) >= nquads || pos >= C_out) return;      // whole waves leave: no barrier below
    const int cl = order ? order[pos] : pos;
    const int g0 = s * CSR_SLAB;
#pragma unroll
    for (int i = 0; i < CSR_SLAB / N / 64; ++i) {
        V z;
        T *zp = reinterpret_cast<T *>(&z);
#pragma unroll
        for (int k = 0; k < N; ++k) zp[k] = T(0);
        reinterpret_cast<V *>(slab)[lane + 64 * i] = z;
    }
    const int64_t p0 = g_indptr[cl], p1 = g_indptr[cl + 1];
    for (int64_t pb = p0; pb < p1; pb += 64) {
        // lane l holds graph entry pb + l: source row, weight, and the row's segment [a, e) inside this slab
        const int cnt = (int)min((int64_t)64, p1 - pb);
        int64_t a_l = 0;
        int n_l = 0;
        T ws_l = T(0);
        if (lane < cnt) {
            const int j = g_indices[pb + lane];
            ws_l = w[pb + lane] * (T)scale[j];
            const int32_t *sp = slabptr + (int64_t)j * (nslab + 1) + s;
            const int o0 = sp[0], o1 = sp[1];
            a_l = indptr[j] + o0;
            n_l = o1 - o0;
        }
        for (int u = 0; u < cnt; ++u) {
            const int64_t a = ((int64_t)__builtin_amdgcn_readlane((int)(a_l >> 32), u) << 32) | (unsigned)__builtin_amdgcn_readlane((int)a_l, u);
            const int n = __builtin_amdgcn_readlane(n_l, u);
            const T ws = readlane_t(ws_l, u);
            for (int t0 = 0; t0 < n; t0 += 64 * CSR_NZ) {
                int g[CSR_NZ];
                CT x[CSR_NZ];
#pragma unroll
                for (int q = 0; q < CSR_NZ; ++q) {
                    const int t = t0 + lane + 64 * q;
                    g[q] = -1;
                    if (t < n) { g[q] = indices[a + t] - g0; x[q] = data[a + t]; }
                }
#pragma unroll
                for (int q = 0; q < CSR_NZ; ++q)
                    if (g[q] >= 0) slab[g[q]] = fma(ws, (T)x[q], slab[g[q]]);
            }
        }
    }
    if (maximum) {                                       // np.maximum(S_sz, Sx): the cell's own scaled counts (analysis.py:1017-1019)
        const int64_t j = cell0 + cl;
        const int32_t *sp = slabptr + j * (nslab + 1) + s;
        const int64_t a = indptr[j] + sp[0];
        const int n = sp[1] - sp[0];
        const T fs = (T)scale[j];
        for (int t = lane; t < n; t += 64) {
            const int g = indices[a + t] - g0;
            const T o = fs * (T)data[a + t];
            slab[g] = slab[g] > o ? slab[g] : o;
        }
    }
    // the dense row piece: 16-byte stores; rows are padded to ld_out (zeros beyond G: no non-zero lives there)
    T *orow = out + (int64_t)cl * ld_out + g0;
    const int nv = (int)min((int64_t)CSR_SLAB, ld_out - g0) / N;
#pragma unroll
    for (int i = 0; i < CSR_SLAB / N / 64; ++i) {
        const int v = lane + 64 * i;
        if (v < nv) reinterpret_cast<V *>(orow)[v] = reinterpret_cast<const V *>(slab)[v];
    }
}

}  // namespace vcy

using namespace vcy;

extern "C" int64_t vcy_csr_slab_genes(void) { return CSR_SLAB; }

extern "C" int vcy_csr_slab_ptr(const int64_t *indptr, const int32_t *indices, int32_t *slabptr, int64_t C, int64_t G, vcy_stream stream)
{
    VCY_REQUIRE(indptr && slabptr && C > 0 && G > 0, "csr_slab_ptr: bad arguments");
    const int64_t nslab = (G + CSR_SLAB - 1) / CSR_SLAB, total = C * (nslab + 1);
    VCY_REQUIRE((total + 255) / 256 < (1LL << 31), "csr_slab_ptr: grid too large");
    hipLaunchKernelGGL(k_csr_slab_ptr, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), indptr, indices, slabptr, C, (int)nslab, CSR_SLAB);
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}

extern "C" int vcy_knn_pool_csr(const int64_t *indptr, const int32_t *indices, const void *data, const int32_t *slabptr, const double *scale,
                                void *out, const int64_t *g_indptr, const int32_t *g_indices, const void *w, const int32_t *order, int64_t C,
                                int64_t G, int64_t ld_out, int64_t cell0, int64_t C_out, int maximum, int count_dtype, int dtype, vcy_stream stream)
{
    VCY_REQUIRE(indptr && indices && data && slabptr && scale && out && g_indptr && g_indices && w, "knn_pool_csr: null pointer");
    VCY_REQUIRE(C > 0 && G > 0 && ld_out >= G && C_out > 0 && cell0 >= 0 && cell0 + C_out <= C, "knn_pool_csr: bad shape");
    VCY_REQUIRE(count_dtype == VCY_U16 || count_dtype == VCY_U8, "knn_pool_csr: count_dtype must be VCY_U16 or VCY_U8");
    VCY_REQUIRE(dtype == VCY_F32 || dtype == VCY_F64, "knn_pool_csr: bad dtype");
    VCY_REQUIRE(ld_out % 16 == 0 && ((uintptr_t)out % 16) == 0, "knn_pool_csr: output rows must be 16-byte aligned with ld_out % 16 == 0");
    const int64_t nslab = (G + CSR_SLAB - 1) / CSR_SLAB;
    const int64_t nquads = (C_out + CSR_WAVES - 1) / CSR_WAVES, blocks = nslab * ((nquads + 7) / 8 * 8);
    VCY_REQUIRE(blocks < (1LL << 31), "knn_pool_csr: grid too large");
    hipStream_t st = as_stream(stream);
#define VCY_POOLCSR(T, CT)                                                                                                                          \
    hipLaunchKernelGGL((k_knn_pool_csr<T, CT>), dim3((unsigned)blocks), dim3(64 * CSR_WAVES), 0, st, indptr, indices, (const CT *)data, slabptr, scale, \
                       (T *)out, g_indptr, g_indices, (const T *)w, order, (int)G, ld_out, cell0, (int)C_out, (int)nslab, maximum)
    if (dtype == VCY_F32) { if (count_dtype == VCY_U16) VCY_POOLCSR(float, uint16_t); else VCY_POOLCSR(float, uint8_t); }
    else { if (count_dtype == VCY_U16) VCY_POOLCSR(double, uint16_t); else VCY_POOLCSR(double, uint8_t); }
#undef VCY_POOLCSR
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}
