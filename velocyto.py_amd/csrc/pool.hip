// pool.hip -- stage A: kNN pooling of the count matrices (the data @ w.T of
// neighbors.convolve_by_sparse_weights, neighbors.py:416-423; analysis.py:1011-1019).
//
// out[c,:] = sum_p w[p] * data[indices[p],:]: a gather-average of k+1 neighbour gene vectors.
// Every cell's vector is gathered by ~k other cells: (k+1) * G * s bytes of gathers per cell against an
// algorithmic 2 * G * s (read once, write once).  The launch therefore walks GENE SLABS, slab-major, and
// inside a slab the cells are scheduled in a locality-sorted order with an XCD-aware block mapping, so that
// co-resident workgroups - which share neighbours - re-read the same 2 KB row pieces from the per-XCD L2
// (measured at 50k x 30k, k = 30: 85 % of the gathers hit L2; the kernel ends up L2-bandwidth-bound at
// ~18 TB/s of gathers, which is why the count-matrix variant below, moving 2-byte elements, is the default).
// The adjacency has k/C ~ 0.06 % density with scattered columns, so this is not a dense block contraction
// and MFMA is deliberately not used (it would multiply by zeros).
#include "common.h"

namespace vcy {

// DUAL: a second matrix pooled with the same weights (data2 -> out2).  W2: ONE matrix pooled with two weight sets over the same
// graph (w -> out, w2 -> out2; calculate_embedding_shift's real and randomised transition probabilities, analysis.py:1716, 1728):
// the rows are gathered once.
template <typename T, bool DUAL, bool W2 = false>
__global__ __launch_bounds__(256) void k_knn_pool(const T *__restrict__ data, T *__restrict__ out, const T *__restrict__ data2,
                                                   T *__restrict__ out2, const int64_t *__restrict__ indptr,
                                                   const int32_t *__restrict__ indices, const T *__restrict__ w, const T *__restrict__ w2,
                                                   const int32_t *__restrict__ order, int G, int64_t ld,
                                                   int64_t cell0, int C_out, int slab, int maximum)
{
    static_assert(!(DUAL && W2), "one extension at a time");
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
        T acc[N], acc2[N];
#pragma unroll
        for (int k = 0; k < N; ++k) { acc[k] = T(0); acc2[k] = T(0); }
        int64_t p = p0;
        for (; p + 3 < p1; p += 4) {                       // 4 (DUAL: 8) gathers in flight
            V x[4], y[4]; T ww[4], wv2[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t ro = (int64_t)indices[p + u] * ld + g0;
                x[u] = reinterpret_cast<const V *>(data + ro)[v];
                if (DUAL) y[u] = reinterpret_cast<const V *>(data2 + ro)[v];
                ww[u] = w[p + u];
                if (W2) wv2[u] = w2[p + u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const T *xp = reinterpret_cast<const T *>(&x[u]);
                const T *yp = reinterpret_cast<const T *>(&y[u]);
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    acc[k] = fma(ww[u], xp[k], acc[k]);
                    if (DUAL) acc2[k] = fma(ww[u], yp[k], acc2[k]);
                    if (W2) acc2[k] = fma(wv2[u], xp[k], acc2[k]);
                }
            }
        }
        for (; p < p1; ++p) {
            const int64_t ro = (int64_t)indices[p] * ld + g0;
            const V xv = reinterpret_cast<const V *>(data + ro)[v];
            V yv;
            if (DUAL) yv = reinterpret_cast<const V *>(data2 + ro)[v];
            const T wv = w[p], wq = W2 ? w2[p] : T(0);
            const T *xp = reinterpret_cast<const T *>(&xv);
            const T *yp = reinterpret_cast<const T *>(&yv);
#pragma unroll
            for (int k = 0; k < N; ++k) {
                acc[k] = fma(wv, xp[k], acc[k]);
                if (DUAL) acc2[k] = fma(wv, yp[k], acc2[k]);
                if (W2) acc2[k] = fma(wq, xp[k], acc2[k]);
            }
        }
        if (maximum) {
            const int64_t ro = (cell0 + cl) * ld + g0;
            const V sv = reinterpret_cast<const V *>(data + ro)[v];
            const T *sp = reinterpret_cast<const T *>(&sv);
#pragma unroll
            for (int k = 0; k < N; ++k) acc[k] = acc[k] > sp[k] ? acc[k] : sp[k];
            if (DUAL) {
                const V tv = reinterpret_cast<const V *>(data2 + ro)[v];
                const T *tp = reinterpret_cast<const T *>(&tv);
#pragma unroll
                for (int k = 0; k < N; ++k) acc2[k] = acc2[k] > tp[k] ? acc2[k] : tp[k];
            }
        }
        V o, o2;
        T *op = reinterpret_cast<T *>(&o), *op2 = reinterpret_cast<T *>(&o2);
#pragma unroll
        for (int k = 0; k < N; ++k) { op[k] = acc[k]; op2[k] = acc2[k]; }
        reinterpret_cast<V *>(out + (int64_t)cl * ld + g0)[v] = o;
        if (DUAL || W2) reinterpret_cast<V *>(out2 + (int64_t)cl * ld + g0)[v] = o2;
    }
    for (int g = g0 + nvec * N + threadIdx.x; g < g1; g += blockDim.x) {   // < N tail genes of the last slab
        T a = T(0), a2 = T(0);
        for (int64_t p = p0; p < p1; ++p) {
            a = fma(w[p], data[(int64_t)indices[p] * ld + g], a);
            if (DUAL) a2 = fma(w[p], data2[(int64_t)indices[p] * ld + g], a2);
            if (W2) a2 = fma(w2[p], data[(int64_t)indices[p] * ld + g], a2);
        }
        if (maximum) {
            const T sv = data[(cell0 + cl) * ld + g]; a = a > sv ? a : sv;
            if (DUAL) { const T tv = data2[(cell0 + cl) * ld + g]; a2 = a2 > tv ? a2 : tv; }
        }
        out[(int64_t)cl * ld + g] = a;
        if (DUAL || W2) out2[(int64_t)cl * ld + g] = a2;
    }
}
}  // namespace vcy

using namespace vcy;

namespace vcy {
// ---------------------------------------------------------------------------------------------
// Count-matrix variant.  The loom layers are uint16 molecule counts (velocyto/constants.py:11) and the
// size-normalised matrices are just  S_sz[c,:] = norm_factor[c] * S[c,:]  (analysis.py:546-549, 573-579), so
// the pooled matrices can be gathered straight from the 2-byte counts,
//     out[c,:] = sum_p (w[p] * scale[indices[p]]) * counts[indices[p],:],
// which halves the bytes every gather moves through L2/HBM.  Thread = 8 genes (one 16-byte load of uint16).
// one 16-byte gather of a count row: 8 uint16 or 16 uint8 genes (layers whose counts all fit a byte are kept as bytes)
template <typename CT> struct alignas(16) CountVec { CT v[16 / sizeof(CT)]; };

template <typename T, typename CT, bool DUAL>
__global__ __launch_bounds__(256) void k_knn_pool_counts(const CT *__restrict__ cS, const CT *__restrict__ cU,
                                                          const double *__restrict__ scaleS, const double *__restrict__ scaleU,
                                                          T *__restrict__ out, T *__restrict__ out2, const int64_t *__restrict__ indptr,
                                                          const int32_t *__restrict__ indices, const T *__restrict__ w,
                                                          const int32_t *__restrict__ order, int G, int64_t ld16, int64_t ld_out,
                                                          int64_t cell0, int C_out, int slab, int maximum)
{
    constexpr int NE = 16 / sizeof(CT);                       // genes per 16-byte gather
    using CV = CountVec<CT>;
    const int per = (C_out + 7) / 8, nblk = per * 8;          // XCD-aware slab-major schedule, as in k_knn_pool
    const int64_t b = blockIdx.x;
    const int s = (int)(b / nblk), bi = (int)(b % nblk);
    const int pos = (bi & 7) * per + (bi >> 3);
    if (pos >= C_out) return;
    const int cl = order ? order[pos] : pos;
    const int g0 = s * slab, g1 = min(G, g0 + slab);
    const int64_t p0 = indptr[cl], p1 = indptr[cl + 1];
    const int nvec = (g1 - g0 + NE - 1) / NE;                 // rows are zero-padded to ld16 (multiple of 64 elements)
    for (int v = threadIdx.x; v < nvec; v += blockDim.x) {
        alignas(16) T acc[NE], acc2[NE];
#pragma unroll
        for (int k = 0; k < NE; ++k) { acc[k] = T(0); acc2[k] = T(0); }
        int64_t p = p0;
        for (; p + 3 < p1; p += 4) {
            CV x[4], y[4]; T ws[4], wu[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = indices[p + u];
                const int64_t ro = (int64_t)j * ld16 + g0;
                x[u] = reinterpret_cast<const CV *>(cS + ro)[v];
                if (DUAL) y[u] = reinterpret_cast<const CV *>(cU + ro)[v];
                const T wp = w[p + u];
                ws[u] = wp * (T)scaleS[j];
                if (DUAL) wu[u] = wp * (T)scaleU[j];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < NE; ++k) {
                    acc[k] = fma(ws[u], (T)x[u].v[k], acc[k]);
                    if (DUAL) acc2[k] = fma(wu[u], (T)y[u].v[k], acc2[k]);
                }
        }
        for (; p < p1; ++p) {
            const int j = indices[p];
            const int64_t ro = (int64_t)j * ld16 + g0;
            const CV xv = reinterpret_cast<const CV *>(cS + ro)[v];
            CV yv;
            if (DUAL) yv = reinterpret_cast<const CV *>(cU + ro)[v];
            const T wp = w[p], wsv = wp * (T)scaleS[j], wuv = DUAL ? wp * (T)scaleU[j] : T(0);
#pragma unroll
            for (int k = 0; k < NE; ++k) { acc[k] = fma(wsv, (T)xv.v[k], acc[k]); if (DUAL) acc2[k] = fma(wuv, (T)yv.v[k], acc2[k]); }
        }
        if (maximum) {
            const int64_t ro = (cell0 + cl) * ld16 + g0;
            const CV sv = reinterpret_cast<const CV *>(cS + ro)[v];
            const T fs = (T)scaleS[cell0 + cl];
#pragma unroll
            for (int k = 0; k < NE; ++k) { const T o = fs * (T)sv.v[k]; acc[k] = acc[k] > o ? acc[k] : o; }
            if (DUAL) {
                const CV tv = reinterpret_cast<const CV *>(cU + ro)[v];
                const T fu = (T)scaleU[cell0 + cl];
#pragma unroll
                for (int k = 0; k < NE; ++k) { const T o = fu * (T)tv.v[k]; acc2[k] = acc2[k] > o ? acc2[k] : o; }
            }
        }
        // 16-byte stores (a lane's NE outputs are contiguous): element-wise 4-byte stores at a 32/64-byte lane stride made this
        // kernel write-bound (13.5 of 15 ms with a single neighbour per cell).  (Round 2: turning the wave's 64 x NE block
        // through LDS so that every store instruction covers a contiguous 1 KiB measured SLOWER - uint8 8.1 -> 10.1 ms,
        // uint16 10.9 -> 15.5 ms: the 64-byte lane stride already fills whole lines over a lane's four stores.)  Output rows are padded to a multiple of 64
        // elements, so a vector that starts inside a row ends inside it; elements past G are written as zeros.
        // (Round 4, same experiment on the kernel as it is now, XOR-swizzled and conflict-free both ways: the ONE-neighbour floor
        // drops - f32 4.3 -> 3.0 ms, f64 10.1 -> 5.5 ms for both layers - but the 31-neighbour kernel does not move, f32 8.02 -> 8.03,
        // f64 13.8 -> 14.0 ms: with 31 gathers per output the stores hide behind the instruction stream.  Counters of the launch,
        // profiles/r04_pool_pmc.txt: 8.0 resident waves per SIMD, a wave issues 14 % of its time, waits to issue 43 % and is
        // parked at s_waitcnt 43 %; 1.5e9 VALU (2.1 per gathered element: v_cvt_f32_ubyte + half a v_pk_fma_f32) + 0.7e9 SALU
        // (the 64-bit row addresses, 15 per neighbour) + 4.7e7 gathers per layer = 3.8 clocks per issued instruction and SIMD.
        // Nontemporal stores: 8.0 -> 10.7 ms (f64: 13.8 -> 28 ms).  Two groups of four gathers in flight, the loads of group i + 1 issued
        // before the arithmetic of group i: 110 / 158 VGPRs instead of 62 / 66 and slower, f32 7.85 -> 9.25 ms, f64 13.6 -> 15.6 ms,
        // profiles/r04b_pool_dbuf.txt - the launch is not waiting for its gathers, resident waves cover them.  Eight waves per SIMD for the
        // f64 / uint8 instance (64 VGPRs instead of 66, 12 bytes of scratch): 13.85 -> 14.32 ms.)
        using OV = typename Vec<T>::type;
        constexpr int ON = Vec<T>::N;
        if (g0 + v * NE < ld_out) {
#pragma unroll
            for (int k = 0; k < NE; ++k) { if (g0 + v * NE + k >= G) { acc[k] = T(0); acc2[k] = T(0); } }
            OV *o1 = reinterpret_cast<OV *>(out + (int64_t)cl * ld_out + g0 + v * NE);
#pragma unroll
            for (int q = 0; q < NE / ON; ++q) o1[q] = *reinterpret_cast<const OV *>(&acc[q * ON]);
            if (DUAL) {
                OV *o2 = reinterpret_cast<OV *>(out2 + (int64_t)cl * ld_out + g0 + v * NE);
#pragma unroll
                for (int q = 0; q < NE / ON; ++q) o2[q] = *reinterpret_cast<const OV *>(&acc2[q * ON]);
            }
        }
    }
}

}  // namespace vcy

static int knn_pool_impl(const void *data, void *out, const void *data2, void *out2, const int64_t *indptr, const int32_t *indices,
                         const void *w, const void *w2, const int32_t *order, int64_t C, int64_t G, int64_t ld, int64_t cell0, int64_t C_out, int maximum,
                         int64_t slab_genes, int dtype, vcy_stream stream)
{
    VCY_REQUIRE(data && out && indptr && indices && w, "knn_pool: null pointer");
    VCY_REQUIRE(w2 ? (data2 == nullptr && out2 != nullptr) : ((data2 == nullptr) == (out2 == nullptr)), "knn_pool: data2/out2 (or w2/out2) go together");
    VCY_REQUIRE(!(w2 && maximum), "knn_pool: maximum is not defined for two weight sets");
    VCY_REQUIRE(C > 0 && G > 0 && ld >= G && C_out > 0 && cell0 >= 0 && cell0 + C_out <= C, "knn_pool: bad shape");
    VCY_REQUIRE(dtype == VCY_F32 || dtype == VCY_F64, "knn_pool: bad dtype");
    const int N = dtype == VCY_F32 ? 4 : 2;
    VCY_REQUIRE(ld % N == 0, "knn_pool: ld must keep rows 16-byte aligned");
    VCY_REQUIRE(data != out && (data2 == nullptr || data2 != out2), "knn_pool: in-place pooling is not supported");
    int64_t slab = slab_genes > 0 ? slab_genes : 512;
    slab = (slab + N - 1) / N * N;
    if (slab > G) slab = (G + N - 1) / N * N;
    const int64_t nslab = (G + slab - 1) / slab;
    const int threads = slab / N >= 256 ? 256 : (slab / N >= 128 ? 128 : 64);
    const int64_t blocks = nslab * ((C_out + 7) / 8 * 8);
    VCY_REQUIRE(blocks < (1LL << 31), "knn_pool: grid too large");
    hipStream_t st = as_stream(stream);
#define VCY_POOL(T, DUAL, W2)                                                                                                      \
    hipLaunchKernelGGL((k_knn_pool<T, DUAL, W2>), dim3((unsigned)blocks), dim3(threads), 0, st, (const T *)data, (T *)out, (const T *)data2, \
                       (T *)out2, indptr, indices, (const T *)w, (const T *)w2, order, (int)G, ld, cell0, (int)C_out, (int)slab, maximum)
    if (dtype == VCY_F32) { if (w2) VCY_POOL(float, false, true); else if (data2) VCY_POOL(float, true, false); else VCY_POOL(float, false, false); }
    else { if (w2) VCY_POOL(double, false, true); else if (data2) VCY_POOL(double, true, false); else VCY_POOL(double, false, false); }
#undef VCY_POOL
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}

extern "C" int vcy_knn_pool(const void *data, void *out, const int64_t *indptr, const int32_t *indices, const void *w,
                            const int32_t *order, int64_t C, int64_t G, int64_t ld, int64_t cell0, int64_t C_out, int maximum,
                            int64_t slab_genes, int dtype, vcy_stream stream)
{
    return knn_pool_impl(data, out, nullptr, nullptr, indptr, indices, w, nullptr, order, C, G, ld, cell0, C_out, maximum, slab_genes, dtype, stream);
}

extern "C" int vcy_knn_pool2(const void *data, void *out, const void *data2, void *out2, const int64_t *indptr, const int32_t *indices,
                             const void *w, const int32_t *order, int64_t C, int64_t G, int64_t ld, int64_t cell0, int64_t C_out,
                             int maximum, int64_t slab_genes, int dtype, vcy_stream stream)
{
    VCY_REQUIRE(data2 && out2, "knn_pool2: null pointer");
    // one launch per matrix: sharing the index / weight reads does not pay for the doubled accumulators (f32, 50k x 30k:
    // 20.4 ms in one launch, 18.5 ms in two; tools/bench_pool.py)
    const int rc = knn_pool_impl(data, out, nullptr, nullptr, indptr, indices, w, nullptr, order, C, G, ld, cell0, C_out, maximum, slab_genes, dtype, stream);
    if (rc) return rc;
    return knn_pool_impl(data2, out2, nullptr, nullptr, indptr, indices, w, nullptr, order, C, G, ld, cell0, C_out, maximum, slab_genes, dtype, stream);
}

extern "C" int vcy_knn_pool_w2(const void *data, void *out, void *out2, const int64_t *indptr, const int32_t *indices, const void *w,
                               const void *w2, const int32_t *order, int64_t C, int64_t G, int64_t ld, int64_t cell0, int64_t C_out,
                               int64_t slab_genes, int dtype, vcy_stream stream)
{
    VCY_REQUIRE(w2 && out2 && out != out2, "knn_pool_w2: null pointer");
    return knn_pool_impl(data, out, nullptr, out2, indptr, indices, w, w2, order, C, G, ld, cell0, C_out, 0, slab_genes, dtype, stream);
}

extern "C" int vcy_knn_pool_counts(const void *countsS, const void *countsU, const double *scaleS, const double *scaleU, void *out,
                                   void *out2, const int64_t *indptr, const int32_t *indices, const void *w, const int32_t *order,
                                   int64_t C, int64_t G, int64_t ld16, int64_t ld_out, int64_t cell0, int64_t C_out, int maximum,
                                   int64_t slab_genes, int count_dtype, int dtype, vcy_stream stream)
{
    VCY_REQUIRE(countsS && scaleS && out && indptr && indices && w, "knn_pool_counts: null pointer");
    VCY_REQUIRE((countsU == nullptr) == (out2 == nullptr) && (countsU == nullptr) == (scaleU == nullptr), "knn_pool_counts: countsU/scaleU/out2 go together");
    VCY_REQUIRE(C > 0 && G > 0 && ld16 >= G && ld_out >= G && C_out > 0 && cell0 >= 0 && cell0 + C_out <= C, "knn_pool_counts: bad shape");
    VCY_REQUIRE(count_dtype == VCY_U16 || count_dtype == VCY_U8, "knn_pool_counts: count_dtype must be VCY_U16 or VCY_U8");
    const int ne = count_dtype == VCY_U16 ? 8 : 16;               // genes per 16-byte gather
    VCY_REQUIRE(ld16 % 16 == 0 && ((uintptr_t)countsS % 16) == 0 && ((uintptr_t)countsU % 16) == 0, "knn_pool_counts: count rows must be 16-byte aligned (ld % 16 == 0)");
    VCY_REQUIRE(dtype == VCY_F32 || dtype == VCY_F64, "knn_pool_counts: bad dtype");
    VCY_REQUIRE(ld_out % 16 == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)out2 % 16) == 0, "knn_pool_counts: output rows must be 16-byte aligned with ld_out % 16 == 0");
    int64_t slab = slab_genes > 0 ? slab_genes : 1024;
    slab = (slab + ne - 1) / ne * ne;
    if (slab > G) slab = (G + ne - 1) / ne * ne;
    const int64_t nslab = (G + slab - 1) / slab;
    const int threads = slab / ne >= 256 ? 256 : (slab / ne >= 128 ? 128 : 64);
    const int64_t blocks = nslab * ((C_out + 7) / 8 * 8);
    VCY_REQUIRE(blocks < (1LL << 31), "knn_pool_counts: grid too large");
    hipStream_t st = as_stream(stream);
    // (Round 2, measured and not kept: one wave per (cell, 64 x 16-byte slab) that reads all its neighbour indices and weights with
    //  two vector loads, hands them out by v_readlane and keeps a ring of 5 row gathers in flight - no scalar-load chain per group of
    //  four neighbours - ran the uint8 layers in 9.0 instead of 7.9 ms and the uint16 layers in 11.5 instead of 10.9: the kernel is
    //  not waiting on that chain but on the L2 -> CU path, 93 GB of row gathers per pass = 11.6 TB/s at an 88 % L2 hit rate.)
    // one launch per layer: pooling both layers in one launch shares the index / weight reads but doubles the accumulators
    // (uint8: 130 VGPRs, 3 waves per SIMD) and measured slower at 50k x 30k - uint16 11.9 vs 11.0 ms, uint8 9.6 vs 8.4 ms
#define VCY_POOLC(T, CT)                                                                                                                       \
    do {                                                                                                                                       \
        hipLaunchKernelGGL((k_knn_pool_counts<T, CT, false>), dim3((unsigned)blocks), dim3(threads), 0, st, (const CT *)countsS,               \
                           (const CT *)nullptr, scaleS, (const double *)nullptr, (T *)out, (T *)nullptr, indptr, indices, (const T *)w, order,  \
                           (int)G, ld16, ld_out, cell0, (int)C_out, (int)slab, maximum);                                                       \
        if (countsU) {                                                                                                                         \
            VCY_LAUNCH_CHECK();                                                                                                                \
            hipLaunchKernelGGL((k_knn_pool_counts<T, CT, false>), dim3((unsigned)blocks), dim3(threads), 0, st, (const CT *)countsU,           \
                               (const CT *)nullptr, scaleU, (const double *)nullptr, (T *)out2, (T *)nullptr, indptr, indices, (const T *)w,   \
                               order, (int)G, ld16, ld_out, cell0, (int)C_out, (int)slab, maximum);                                            \
        }                                                                                                                                      \
    } while (0)
    if (dtype == VCY_F32) { if (count_dtype == VCY_U16) VCY_POOLC(float, uint16_t); else VCY_POOLC(float, uint8_t); }
    else { if (count_dtype == VCY_U16) VCY_POOLC(double, uint16_t); else VCY_POOLC(double, uint8_t); }
#undef VCY_POOLC
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}
