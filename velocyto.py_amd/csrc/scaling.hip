// scaling.hip -- stage E: the expression scaling of calculate_embedding_shift without its (genes, cells) temporaries.
//
// Reference (velocyto/analysis.py:1714-1719, and :1726-1731 for the randomised control):
//     estim_delta = hi_dim @ transition_prob.T - hi_dim @ (embedding_knn / n).T          (genes, cells): two dense x (C, C) products
//     cos_proj    = (delta_S * estim_delta).sum(0) / sqrt((estim_delta ** 2).sum(0))      (cells)
//     scaling     = clip(cos_proj / scaling_penalty, 0, 1)
// In neighbour-list form estim_delta[:, c] = sum_k wdiff[c, k] hi_dim[:, ixs[c, k]] with wdiff = tp - 1/n (vcy_transition_prob):
// the pooling of a cell's n sampled neighbours' gene vectors - C * n * G multiply-adds, as many pair-genes as stage D evaluates.
// Until round 3 this ran as vcy_knn_pool_w2 (real and control weights over one gather) followed by vcy_row_cosproj: every
// (cell, neighbour) pair pulled the neighbour's whole row through the CU's vector-memory path (250 x 120 KB per cell: the launch
// was bound by that path, 64 ms in f32 and 131 ms in f64 at 50 000 x 30 000), and the two estimates were written out (12 / 24 GB)
// only to be reduced to one number per cell.
//
// Here a workgroup owns GC = 8 cells that are adjacent in the schedule order (the Hilbert curve of the embedding: they share most of
// their sampled neighbours, 3.5 x on the bench workload) and walks the UNION of their neighbour lists, exactly as stage D does
// (csrc/coldeltacor.hip): the (neighbour, member, slot) keys are bitonic-sorted in LDS, a run of equal neighbours is a ROW with one
// descriptor (neighbour, member mask, first pair), the pairs' weights are gathered into LDS in sorted order.  Each wave then owns
// every 8th chunk of 64 x 16 bytes of the gene axis: it loads a row's chunk ONCE (the next row's load in flight) and adds it, times
// the member's weight(s) (f32: v_readlane; f64: read in place through DPP row_newbcast, 95.7 -> 89.4 ms), into the register accumulators of every member that lists it - estim_delta[member] for that chunk, for the
// real and the control weights.  After the last row the chunk of delta_S (and delta_S_rndm) of every member is read once and the
// two sums of :1717 are folded in fp64; estim_delta never leaves the registers.  Per member the neighbours are added in ascending
// cell number; sums over genes and waves are folded in a fixed order: results are reproducible run to run.
// Roofline: the VALU (two FMAs per weight set, element and (cell, neighbour) pair: 2 * C * n * G flop per set); the row gathers
// drop by the sharing factor of the groups.
#include "common.h"

namespace vcy {

constexpr int SC_GC = 8;             // cells per workgroup
constexpr int SC_WAVES = 8;          // waves per workgroup (512 threads; two workgroups per CU)
constexpr int SC_MAXN = 256;         // widest neighbour list one workgroup sorts (GC * n <= 2048 pairs)

template <typename T> struct W2 { T a, b; };

// acc += w(lane M of the own 16-lane row) * x: the f64 multiply-add reads its first operand through DPP row_newbcast (the one DPP
// control the double-precision ALU has), so a member's weight needs no v_readlane and no SGPR - it only has to sit in lane M of every row
// (A DPP read of a VGPR needs two wait states after a VALU write of it and the compiler does not see inside an asm statement: the weights
// come straight from an LDS read; tests/test_host_and_abi.py checks the listing for a VALU write of the operand in the two slots before.)
template <int M> __device__ __forceinline__ void fmac_bcast(double &acc, double w, double x)
{
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "v"(x), "n"(M));
}

template <typename T> __host__ __device__ inline size_t scaling_lds_bytes(int npad, int maxpairs)
{
    return (size_t)npad * 8 + (size_t)(maxpairs + 2) * 8 + (size_t)maxpairs * sizeof(W2<T>) + (size_t)SC_WAVES * SC_GC * 4 * sizeof(double) + 64 * sizeof(int);
}

// DUAL: a second weight set w2 / second matrix dS2 (the randomised control) over the same lists
template <typename T, bool DUAL>
__global__ __launch_bounds__(64 * SC_WAVES, 4) void k_embedding_scaling(const T *__restrict__ hi, const T *__restrict__ dS, const T *__restrict__ dS2,
                                                                         const int32_t *__restrict__ ixs, const T *__restrict__ w, const T *__restrict__ w2,
                                                                         const int32_t *__restrict__ order, double *__restrict__ cos1, double *__restrict__ cos2,
                                                                         int G, int64_t ld, int C_out, int n, int npad)
{
    using V = typename Vec<T>::type;
    constexpr int N = Vec<T>::N;
    constexpr bool BCAST = sizeof(T) == 8;                           // weights reach the f64 multiply-adds through DPP (fmac_bcast)
    static_assert(SC_GC == 8, "fmac_bcast is instantiated for members 0..7");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int maxpairs = SC_GC * n;
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(smem);                       // [npad]
    unsigned long long *desc = keys + npad;                                                         // [maxpairs + 2] row descriptors
    W2<T> *wp = reinterpret_cast<W2<T> *>(desc + maxpairs + 2);                                     // [maxpairs] weights in sorted pair order
    double *sums = reinterpret_cast<double *>(wp + maxpairs);                                       // [waves][GC][4] num1, den1, num2, den2
    int *s_cells = reinterpret_cast<int *>(sums + SC_WAVES * SC_GC * 4);                            // [GC]
    int *s_wavetot = s_cells + SC_GC;                                                               // [SC_WAVES]
    int &s_U = s_wavetot[SC_WAVES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware schedule (as stage D): workgroup b runs on XCD b % 8 -> an XCD owns a contiguous range of groups
    const int ngroups = (C_out + SC_GC - 1) / SC_GC, per = (ngroups + 7) / 8;
    const int gpos = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    if (gpos >= ngroups) return;
    const int g0cell = gpos * SC_GC, gcount = min(SC_GC, C_out - g0cell), npairs = gcount * n;
    if (tid < SC_GC) s_cells[tid] = tid < gcount ? (order ? order[g0cell + tid] : g0cell + tid) : 0;
    for (int t = tid; t < SC_WAVES * SC_GC * 4; t += blockDim.x) sums[t] = 0.0;
    __syncthreads();
    // ---- keys = neighbour << 16 | member << 12 | slot, sorted (npad: next power of two >= GC * n)
    for (int t = tid; t < npad; t += blockDim.x) {
        unsigned long long key = ~0ull;
        if (t < npairs) {
            const int m = t / n, k = t - m * n;
            key = ((unsigned long long)(unsigned)ixs[(int64_t)s_cells[m] * n + k] << 16) | ((unsigned long long)m << 12) | (unsigned)k;
        }
        keys[t] = key;
    }
    __syncthreads();
    for (int size = 2; size <= npad; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < npad / 2; t += blockDim.x) {
                const int lo = 2 * t - (t & (stride - 1)), hi_ = lo + stride;
                const bool up = ((lo & size) == 0);
                const unsigned long long a = keys[lo], b = keys[hi_];
                if ((a > b) == up) { keys[lo] = b; keys[hi_] = a; }
            }
            __syncthreads();
        }
    // ---- rows: runs of pairs with the same neighbour (a member listed twice for one neighbour starts a new run);
    //      desc[r] = neighbour << 19 | member mask << 11 | first pair; the pairs' weights in sorted order
    {
        auto head = [&](int t) {
            if (t == 0) return true;
            const unsigned long long a = keys[t] >> 12, b = keys[t - 1] >> 12;
            return (a >> 4) != (b >> 4) || a == b;
        };
        const int perth = (npad + blockDim.x - 1) / blockDim.x;
        const int t0 = tid * perth, t1 = min(npairs, t0 + perth);
        int cnt = 0;
        for (int t = t0; t < t1; ++t) cnt += head(t) ? 1 : 0;
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
        if (lane == 63) s_wavetot[wave] = incl;
        __syncthreads();
        int base = 0;
        for (int q = 0; q < wave; ++q) base += s_wavetot[q];
        int rank = base + incl - cnt;
        for (int t = t0; t < t1; ++t)
            if (head(t)) {
                unsigned mask = 0;
                int q = t;
                do { mask |= 1u << (unsigned)((keys[q] >> 12) & 15); ++q; } while (q < npairs && !head(q));
                desc[rank++] = ((keys[t] >> 16) << 19) | ((unsigned long long)mask << 11) | (unsigned)t;
            }
        if (tid == blockDim.x - 1) s_U = base + incl;
        for (int t = tid; t < npairs; t += blockDim.x) {
            const unsigned long long key = keys[t];
            const int64_t src = (int64_t)s_cells[(int)((key >> 12) & 15)] * n + (int)(key & 4095);
            wp[t] = W2<T>{w[src], DUAL ? w2[src] : T(0)};
        }
        __syncthreads();
    }
    const int U = s_U;
    const int nvec = (G + N - 1) / N;                                  // rows are zero-padded to ld: a partial last vector reads zeros
    // ---- the wave's chunks of the gene axis
    for (int v0 = wave * 64; v0 < nvec; v0 += 64 * SC_WAVES) {
        const int v = v0 + lane;
        const bool in = v < nvec;
        const int64_t voff = (int64_t)(in ? v : 0) * N;
        T acc[SC_GC][N], acc2[SC_GC][N];
#pragma unroll
        for (int m = 0; m < SC_GC; ++m)
#pragma unroll
            for (int k = 0; k < N; ++k) { acc[m][k] = T(0); acc2[m][k] = T(0); }
        // A row costs a wave ~100 clocks of multiply-adds and its gather one to two microseconds: SC_PF (3 / 4) rows are kept in flight per
        // wave (gene chunk, weights, descriptor), and descriptors are requested another SC_PF rows ahead, so that neither a row's load
        // address nor its multiply-adds wait for an LDS or memory round trip issued in the same row.  A row's weights are contiguous
        // in sorted pair order: lanes 0..7 fetch its (at most GC) pairs with one LDS read, a member's weight is then broadcast out of
        // that register by v_readlane (wave-uniform lane number = how many members of the mask came before).
        constexpr int SC_PF = sizeof(T) == 8 ? 3 : 4;               // (f64: 2 / 3 / 4 / 5 rows in flight 105.4 / 84.9 / 88.4 / 95.6 ms; f32: 2 / 3 / 4 50.4 / 46.1 / 44.9)
        auto desc_at = [&](int r) { return desc[min(r, max(U - 1, 0))]; };
        // f32: lanes 0..7 hold the row's pairs in sorted order (the j-th member of the mask in lane j, handed out by v_readlane);
        // f64: lane m of every 16-lane row holds MEMBER m's pair (first pair + members of the mask below m), read through DPP
        auto row_weights = [&](unsigned long long d) {
            int off = lane & 7;
            if (BCAST) off = __builtin_popcount(((unsigned)(d >> 11) & 255u) & ((1u << (lane & 7)) - 1u));
            return wp[min((int)(unsigned)(d & 2047) + off, max(npairs - 1, 0))];
        };
        auto row_chunk = [&](unsigned long long d) { return *reinterpret_cast<const V *>(hi + (int64_t)(d >> 19) * ld + voff); };
        unsigned long long dcur[SC_PF], dnext[SC_PF];
        V xq[SC_PF];
        W2<T> wq[SC_PF];
#pragma unroll
        for (int u = 0; u < SC_PF; ++u) {
            dcur[u] = U > 0 ? desc_at(u) : 0ull;
            dnext[u] = U > 0 ? desc_at(u + SC_PF) : 0ull;
            xq[u] = U > 0 ? row_chunk(dcur[u]) : V{};
            wq[u] = U > 0 ? row_weights(dcur[u]) : W2<T>{T(0), T(0)};
        }
        for (int r0 = 0; r0 < U; r0 += SC_PF) {
#pragma unroll
            for (int u = 0; u < SC_PF; ++u) {
                if (r0 + u < U) {                                      // wave-uniform
                    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)dcur[u]);
                    const unsigned mask = (lo >> 11) & 255u;
                    const T *xp = reinterpret_cast<const T *>(&xq[u]);
                    int j = 0;
                    if constexpr (BCAST) {
#define VCY_SC_MEMBER(M)                                                                                       \
                        if (mask & (1u << M)) {                        /* wave-uniform */                       \
                            _Pragma("unroll") for (int k = 0; k < N; ++k) {                                     \
                                fmac_bcast<M>(acc[M][k], wq[u].a, xp[k]);                                       \
                                if (DUAL) fmac_bcast<M>(acc2[M][k], wq[u].b, xp[k]);                            \
                            }                                                                                   \
                        }
                        VCY_SC_MEMBER(0) VCY_SC_MEMBER(1) VCY_SC_MEMBER(2) VCY_SC_MEMBER(3)
                        VCY_SC_MEMBER(4) VCY_SC_MEMBER(5) VCY_SC_MEMBER(6) VCY_SC_MEMBER(7)
#undef VCY_SC_MEMBER
                        (void)j;
                    } else {
#pragma unroll
                        for (int m = 0; m < SC_GC; ++m) {
                            if (mask & (1u << m)) {                    // wave-uniform
                                const T wa = readlane_t(wq[u].a, j), wb = DUAL ? readlane_t(wq[u].b, j) : T(0);
                                ++j;
#pragma unroll
                                for (int k = 0; k < N; ++k) {
                                    acc[m][k] = fma(wa, xp[k], acc[m][k]);
                                    if (DUAL) acc2[m][k] = fma(wb, xp[k], acc2[m][k]);
                                }
                            }
                        }
                    }
                    // refill the slot with row r + SC_PF (its descriptor was requested SC_PF rows ago), request the one after
                    dcur[u] = dnext[u];
                    xq[u] = row_chunk(dcur[u]);
                    wq[u] = row_weights(dcur[u]);
                    dnext[u] = desc_at(r0 + u + 2 * SC_PF);
                }
            }
        }
        // ---- fold the chunk into the member's sums: sum_g dS * estim, sum_g estim^2 (fp64), one transposing wave reduction per member
#pragma unroll
        for (int m = 0; m < SC_GC; ++m) {
            if (m < gcount) {
                double n1 = 0.0, d1 = 0.0, n2 = 0.0, d2 = 0.0;
                if (in) {
                    const int64_t ro = (int64_t)s_cells[m] * ld + voff;
                    const V a = *reinterpret_cast<const V *>(dS + ro);
                    const T *ap = reinterpret_cast<const T *>(&a);
#pragma unroll
                    for (int k = 0; k < N; ++k) { const double e = (double)acc[m][k]; n1 = fma((double)ap[k], e, n1); d1 = fma(e, e, d1); }
                    if (DUAL) {
                        const V b = *reinterpret_cast<const V *>(dS2 + ro);
                        const T *bp = reinterpret_cast<const T *>(&b);
#pragma unroll
                        for (int k = 0; k < N; ++k) { const double e = (double)acc2[m][k]; n2 = fma((double)bp[k], e, n2); d2 = fma(e, e, d2); }
                    }
                }
                const double tot = wave_sum_rows(n1, d1, n2, d2);     // row r of the wave holds total r
                if ((lane & 15) == 0) sums[(wave * SC_GC + m) * 4 + (lane >> 4)] += tot;           // the wave's own slots: program order
            }
            __builtin_amdgcn_sched_barrier(0);                        // one member at a time: hoisting all 16 row loads costs 64 VGPRs
        }
    }
    __syncthreads();
    if (tid < gcount) {
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        for (int q = 0; q < SC_WAVES; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] += sums[(q * SC_GC + tid) * 4 + j];
        cos1[s_cells[tid]] = s[0] / sqrt(s[1]);                        // 0 / 0 = NaN stays NaN, like the reference's division
        if (DUAL) cos2[s_cells[tid]] = s[2] / sqrt(s[3]);
    }
}

}  // namespace vcy

using namespace vcy;

extern "C" int vcy_embedding_scaling_max_neighbors(void) { return SC_MAXN; }

extern "C" int vcy_embedding_scaling(const void *hi_dim, const void *delta_S, const void *delta_S_rndm, const int32_t *ixs, const void *wdiff,
                                     const void *wdiff_rndm, const int32_t *order, double *cos_proj, double *cos_proj_rndm, int64_t C, int64_t G,
                                     int64_t ld, int64_t C_out, int64_t n, int dtype, vcy_stream stream)
{
    VCY_REQUIRE(hi_dim && delta_S && ixs && wdiff && cos_proj, "embedding_scaling: null pointer");
    VCY_REQUIRE((delta_S_rndm == nullptr) == (wdiff_rndm == nullptr) && (delta_S_rndm == nullptr) == (cos_proj_rndm == nullptr),
                "embedding_scaling: delta_S_rndm / wdiff_rndm / cos_proj_rndm go together");
    VCY_REQUIRE(C > 0 && G > 0 && ld >= G && C_out > 0 && C_out <= C && n > 0, "embedding_scaling: bad shape");
    VCY_REQUIRE(dtype == VCY_F32 || dtype == VCY_F64, "embedding_scaling: bad dtype");
    VCY_REQUIRE(ld % (dtype == VCY_F32 ? 4 : 2) == 0 && ((uintptr_t)hi_dim % 16) == 0 && ((uintptr_t)delta_S % 16) == 0 && ((uintptr_t)delta_S_rndm % 16) == 0,
                "embedding_scaling: rows must be 16-byte aligned");
    if (n > SC_MAXN) return fail(VCY_ERR_UNSUPPORTED, "%s: neighbour lists wider than %lld are pooled with vcy_knn_pool_w2 + vcy_row_cosproj", "embedding_scaling", (long long)SC_MAXN);
    int npad = 2;
    while (npad < SC_GC * n) npad <<= 1;
    const int64_t ngroups = (C_out + SC_GC - 1) / SC_GC, blocks = (ngroups + 7) / 8 * 8;
    hipStream_t st = as_stream(stream);
    const bool dual = delta_S_rndm != nullptr;
#define VCY_SCALING(T, D)                                                                                                                  \
    do {                                                                                                                                   \
        const size_t lds = scaling_lds_bytes<T>(npad, (int)(SC_GC * n));                                                                  \
        auto kern = k_embedding_scaling<T, D>;                                                                                             \
        int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds);                                                            \
        if (rc) return rc;                                                                                                                 \
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * SC_WAVES), lds, st, (const T *)hi_dim, (const T *)delta_S, (const T *)delta_S_rndm, \
                           ixs, (const T *)wdiff, (const T *)wdiff_rndm, order, cos_proj, cos_proj_rndm, (int)G, ld, (int)C_out, (int)n, npad); \
    } while (0)
    if (dtype == VCY_F32) { if (dual) VCY_SCALING(float, true); else VCY_SCALING(float, false); }
    else { if (dual) VCY_SCALING(double, true); else VCY_SCALING(double, false); }
#undef VCY_SCALING
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}
