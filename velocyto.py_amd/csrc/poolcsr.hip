// poolcsr.hip -- stage A at atlas scale: kNN pooling straight from SPARSE (CSR) count layers.
//
// Reference: the same product as pool.hip, data @ w.T of neighbors.convolve_by_sparse_weights (neighbors.py:416-423)
// applied to S_sz = factor * S (analysis.py:546-549, 1011-1016), for datasets whose count layers do not fit as dense
// matrices (BASELINE.json configs[4]: 1M cells x 30k genes at ~8 % density; the reference itself loads every layer dense,
// analysis.py:59-61, which is what caps it).  A cell's row is (gene index int32, count uint8/uint16) pairs, genes ascending.
//
// out[c, :] = sum_p (w[p] * scale[j_p]) * counts[j_p, :]  over the k+1 graph entries p of cell c:  a MERGE of k+1 sparse
// rows into one dense row.  One WAVE owns a cell and walks its gene slabs: the slab (2048 genes, 8 KB of f32) lives in LDS, the
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

// One wave = one output cell x a contiguous range of gene slabs.  Per-entry facts (source row, weight x size factor, row
// start, slab boundary table) are loaded once into lane registers (lane = graph entry) and handed out with v_readlane; slab
// boundaries of consecutive slabs share an endpoint, so each further slab costs one 4-byte load per entry, issued a slab
// ahead.  The non-zeros of CSR_ROWS rows are requested together (up to CSR_NZ x 64 of each: 32 loads in flight per lane)
// and then applied row by row in graph order: one round trip per CSR_ROWS rows instead of one per row.
constexpr int CSR_ROWS = 4;

template <typename T, typename CT>
__global__ __launch_bounds__(64 * CSR_WAVES) void k_knn_pool_csr(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                                  const CT *__restrict__ data, const int32_t *__restrict__ slabptr,
                                                                  const double *__restrict__ scale, T *__restrict__ out,
                                                                  const int64_t *__restrict__ g_indptr, const int32_t *__restrict__ g_indices,
                                                                  const T *__restrict__ w, const int32_t *__restrict__ order, int G, int64_t ld_out,
                                                                  int64_t cell0, int C_out, int nslab, int nsplit, int maximum, int64_t C_rows)
{
    using V = typename Vec<T>::type;
    constexpr int N = Vec<T>::N;
    __shared__ __attribute__((aligned(16))) T lds[CSR_WAVES][CSR_SLAB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T *slab = lds[wave];
    // split-major, XCD-aware schedule as in pool.hip: workgroup b runs on XCD b % 8 (observed; speed only), so XCD x is given a
    // contiguous range of the (locality-sorted) cell schedule and co-resident waves re-read the same CSR rows from one L2
    const int nquads = (C_out + CSR_WAVES - 1) / CSR_WAVES, per = (nquads + 7) / 8, nblk = per * 8;
    const int64_t b = blockIdx.x;
    const int split = (int)(b / nblk), bi = (int)(b % nblk);
    const int quad = (bi & 7) * per + (bi >> 3);
    const int pos = quad * CSR_WAVES + wave;
    if (quad >= nquads || pos >= C_out) return;         // whole waves leave: there is no barrier below
    const int cl = order ? order[pos] : pos;
    const int sper = (nslab + nsplit - 1) / nsplit, s0 = split * sper, s1 = min(nslab, s0 + sper);
    const int64_t p0 = g_indptr[cl], p1 = g_indptr[cl + 1];
    const int64_t own = cell0 + cl;
    // (the 16-byte loads below never read past it; contract of vcy_knn_pool_csr: `indices` / `data` hold at least 4 elements -
    //  a layer with fewer non-zeros is padded by the caller, elements at or past indptr[C] belong to no row and are masked)
    const int64_t nnz_total = max(indptr[C_rows], (int64_t)4);

    for (int64_t pb = p0; pb < p1 || pb == p0; pb += 64) {     // batches of 64 graph entries (one batch for any kNN graph)
        const bool first = pb == p0, last = pb + 64 >= p1;
        const int cnt = (int)max((int64_t)0, min((int64_t)64, p1 - pb));
        int64_t base_l = 0;
        const int32_t *sp_l = slabptr;
        T ws_l = T(0);
        int lo_l = 0, hi_l = 0;
        if (lane < cnt) {
            const int j = g_indices[pb + lane];
            ws_l = w[pb + lane] * (T)scale[j];
            base_l = indptr[j];
            sp_l = slabptr + (int64_t)j * (nslab + 1);
            if (s0 < s1) { lo_l = sp_l[s0]; hi_l = sp_l[s0 + 1]; }
        }
        for (int s = s0; s < s1; ++s) {
            const int g0 = s * CSR_SLAB;
            const int64_t a_l = base_l + lo_l;
            const int n_l = hi_l - lo_l;
            lo_l = hi_l;
            if (lane < cnt && s + 1 < s1) hi_l = sp_l[s + 2];   // next slab's far boundary, a slab ahead
            if (first) {
#pragma unroll
                for (int i = 0; i < CSR_SLAB / N / 64; ++i) {
                    V z;
                    T *zp = reinterpret_cast<T *>(&z);
#pragma unroll
                    for (int k = 0; k < N; ++k) zp[k] = T(0);
                    reinterpret_cast<V *>(slab)[lane + 64 * i] = z;
                }
            } else {                                             // rows of more than 64 entries: continue from the partial sums
                const T *orow = out + (int64_t)cl * ld_out + g0;
                const int nv = (int)min((int64_t)CSR_SLAB, ld_out - g0) / N;
#pragma unroll
                for (int i = 0; i < CSR_SLAB / N / 64; ++i) {
                    const int v = lane + 64 * i;
                    if (v < nv) reinterpret_cast<V *>(slab)[v] = reinterpret_cast<const V *>(orow)[v];
                }
            }
            // The non-zeros of CSR_ROWS rows are requested together and applied row by row.  Lane t takes the FOUR CONSECUTIVE
            // non-zeros 4 t .. 4 t + 3 of a row's segment: ONE 16-byte load of the gene numbers and one load of the counts per row
            // and lane (8 -> 2 vector-memory instructions per row).  The request phase is nothing but loads - no arithmetic on what
            // they return, no branch on it - so that the compiler waits for none of them before the last has been issued (a
            // conditional element-wise tail path made it wait per row: one round trip per row, whatever CSR_ROWS was).  The quad
            // that would run past the end of a segment is shifted back to END with the segment (and, at the very end of the
            // arrays, inside them); the merge phase applies only the elements 4 t <= e < n it owns.
            static_assert(CSR_NZ == 4, "a lane owns one quad of consecutive non-zeros");
            constexpr int XW = (int)sizeof(CT);                  // dwords holding a lane's four counts (uint8: 1, uint16: 2)
            for (int u0 = 0; u0 < cnt; u0 += CSR_ROWS) {
                int gq[CSR_ROWS][4];
                unsigned xw[CSR_ROWS][XW];
                T ws[CSR_ROWS];
                int64_t a[CSR_ROWS];
                int n[CSR_ROWS], sh[CSR_ROWS];
                const int t4 = 4 * lane;
#pragma unroll
                for (int r = 0; r < CSR_ROWS; ++r) {
                    const int u = min(u0 + r, cnt - 1);
                    a[r] = ((int64_t)__builtin_amdgcn_readlane((int)(a_l >> 32), u) << 32) | (unsigned)__builtin_amdgcn_readlane((int)a_l, u);
                    n[r] = (u0 + r < cnt) ? __builtin_amdgcn_readlane(n_l, u) : 0;
                    ws[r] = readlane_t(ws_l, u);
                    // first element of the lane's quad, relative to a[r]: 4 t, pulled back so that the quad ends inside the arrays
                    int s4 = min(t4, max(n[r] - 4, 0));
                    s4 = (int)min((int64_t)s4, nnz_total - 4 - a[r]);
                    sh[r] = s4;
                    if (t4 < n[r]) {
                        __builtin_memcpy(gq[r], indices + a[r] + s4, 16);
                        __builtin_memcpy(xw[r], data + a[r] + s4, 4 * sizeof(CT));
                    }
                }
#pragma unroll
                for (int r = 0; r < CSR_ROWS; ++r) {
                    // the four non-zeros of a lane (and of all lanes) of ONE row are distinct genes: their read-modify-writes do not
                    // alias, so the four reads go out together, then the four updates (LDS executes a wave's operations in order: the
                    // next row's reads see these writes)
                    bool ok[4];
                    int gg[4];
                    T cur[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int el = sh[r] + q;                // element of the segment this quad position holds
                        ok[q] = t4 < n[r] && el >= t4 && el < n[r];
                        gg[q] = ok[q] ? gq[r][q] - g0 : 0;
                        cur[q] = slab[gg[q]];
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const unsigned cv = sizeof(CT) == 1 ? (xw[r][0] >> (8 * q)) & 0xffu : (xw[r][q >> 1] >> (16 * (q & 1))) & 0xffffu;
                        if (ok[q]) slab[gg[q]] = fma(ws[r], (T)cv, cur[q]);
                    }
                    for (int t = 64 * CSR_NZ + lane; t < n[r]; t += 64) {       // segments longer than 256 non-zeros (dense rows)
                        const int g2 = indices[a[r] + t] - g0;
                        slab[g2] = fma(ws[r], (T)data[a[r] + t], slab[g2]);
                    }
                }
            }
            if (maximum && last) {                       // np.maximum(S_sz, Sx): the cell's own scaled counts (analysis.py:1017-1019)
                const int32_t *sp = slabptr + own * (nslab + 1) + s;
                const int64_t ao = indptr[own] + sp[0];
                const int no = sp[1] - sp[0];
                const T fs = (T)scale[own];
                for (int t = lane; t < no; t += 64) {
                    const int gg = indices[ao + t] - g0;
                    const T o = fs * (T)data[ao + t];
                    slab[gg] = slab[gg] > o ? slab[gg] : o;
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
    // a wave walks a contiguous range of slabs of its cell; few output cells (halo rows, small blocks) split the slabs over
    // more waves so that the launch still fills the 256 CUs x 20 resident waves
    int64_t nsplit = (4 * 5120 + C_out - 1) / C_out;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > nslab) nsplit = nslab;
    nsplit = (nslab + ((nslab + nsplit - 1) / nsplit) - 1) / ((nslab + nsplit - 1) / nsplit);      // no empty ranges
    const int64_t nquads = (C_out + CSR_WAVES - 1) / CSR_WAVES, blocks = nsplit * ((nquads + 7) / 8 * 8);
    VCY_REQUIRE(blocks < (1LL << 31), "knn_pool_csr: grid too large");
    hipStream_t st = as_stream(stream);
#define VCY_POOLCSR(T, CT)                                                                                                                          \
    hipLaunchKernelGGL((k_knn_pool_csr<T, CT>), dim3((unsigned)blocks), dim3(64 * CSR_WAVES), 0, st, indptr, indices, (const CT *)data, slabptr, scale, \
                       (T *)out, g_indptr, g_indices, (const T *)w, order, (int)G, ld_out, cell0, (int)C_out, (int)nslab, (int)nsplit, maximum, C)
    if (dtype == VCY_F32) { if (count_dtype == VCY_U16) VCY_POOLCSR(float, uint16_t); else VCY_POOLCSR(float, uint8_t); }
    else { if (count_dtype == VCY_U16) VCY_POOLCSR(double, uint16_t); else VCY_POOLCSR(double, uint8_t); }
#undef VCY_POOLCSR
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}
