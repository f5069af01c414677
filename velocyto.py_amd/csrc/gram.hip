// gram.hip -- the dense contraction next to the path: perform_PCA's covariance product.
//
// Reference: VelocytoLoom.perform_PCA (velocyto/analysis.py:678-702) hands S_norm[pca_genes].T (cells x genes) to
// sklearn.decomposition.PCA, whose arithmetic is: centre every gene, then the spectral decomposition of the centred matrix.
// On the device the matrix is cells-major, X (C, ld); the covariance route needs
//     gram[i][j] = sum_c (X[c][i] - mean[i]) (X[c][j] - mean[j])          (G x G, fp64)
// and the subspace iteration for few components of many genes needs the same contraction against a thin block,
//     out[i][j]  = sum_c (X[c][i] - mean[i]) Y[c][j]                      (G x L, Y fp64 (C, L)).
// Both contract over CELLS, the slow dimension of both operands ("TN" product): a slab of KS cells x 128 genes is one
// contiguous 1 KiB run per cell, staged once into LDS, and every staged f64 feeds 8 matrix instructions; the centring costs one
// VALU subtract per element and no extra pass.  Two forms: k_gram_dma (the one that runs: slabs DMA'd from global memory straight into
// LDS, the mean subtracted at the fragment read) and k_gram (slabs staged through registers, the mean subtracted on the way into LDS;
// VCY_GRAM_DMA=0, kept for A/B).  Measured at 50 000 cells (tools/bench_gram.py, profiles/r05_gram.txt): 10 000 genes 86.5 -> 82.0 ms
// (f64 storage; 0.80 of the f64 matrix peak) and 96.7 -> 79.2 ms (f32 storage, 0.83: the f32 -> f64 converts no longer sit between a load
// and an LDS write); the thin block product streams X at 3.1 TB/s instead of 2.8.
//
// Matrix core use.  v_mfma_f64_16x16x4_f64: D(16x16) += A(16x4) B(4x16), one f64 of A and of B per lane
// (A[row = lane & 15][k = lane >> 4], B[k = lane >> 4][col = lane & 15]), four f64 of D per lane
// (col = lane & 15, row = (lane >> 4) + 4 reg - NOT the f32 map).  Both operands of the TN product are read from LDS tiles laid
// out [cell][gene], so A- and B-fragments are the same access: 16 consecutive doubles of cell k0 + (lane >> 4).
// A workgroup of 4 waves owns a 128 x 128 tile of the output; wave (wm, wn) owns 64 x 64 of it = 4 x 4 MFMA tiles = 64 f64
// accumulators per lane (128 VGPRs); per step of 4 cells it reads 4 + 4 fragments and issues 16 MFMAs (8 LDS bytes per 2048
// flop).  Peak: 2048 flop per instruction at one instruction per 64 clocks per SIMD = 78.6 Tflop/s on 1024 SIMDs at 2.4 GHz - the
// same rate as the f64 vector unit, but without its operand traffic.
// Symmetry: the Gram matrix is computed on the tiles of its upper triangle only (diagonal tiles whole) and mirrored on write.
// Few output tiles (3000 genes: 300) would leave the 256 CUs x 2 resident workgroups unevenly loaded: the cells are then split
// over `ksplit` workgroups per tile, each writing its partial tile to a workspace; k_gram_reduce adds the partials in a fixed
// order (deterministic - no atomics) and mirrors.
#include "common.h"
#include <type_traits>
#ifndef VCY_EXP
#define VCY_EXP 0
#endif

namespace vcy {

typedef double v4d_t __attribute__((ext_vector_type(4)));

constexpr int GM_T = 128;          // edge of a workgroup's output tile
constexpr int GM_KS = 16;          // cells per staged slab
constexpr int GM_LD = GM_T + 16;   // LDS row pitch in doubles: consecutive cells start 32 banks apart
constexpr int GM_THREADS = 256;

// two consecutive elements from an address that is always inside the matrix (callers clamp row and column and mask the values
// afterwards: no branch around a load, so the compiler never closes a load with its own s_waitcnt)
template <typename T> __device__ __forceinline__ void load2(const T *p, double &a, double &b)
{
    if constexpr (sizeof(T) == 8) { const double2 v = *reinterpret_cast<const double2 *>(p); a = v.x; b = v.y; }
    else { const float2 v = *reinterpret_cast<const float2 *>(p); a = (double)v.x; b = (double)v.y; }
}

// ---- write of a wave's accumulators: D[row = (lane >> 4) + 4 reg][col = lane & 15] of every 16 x 16 tile
// (split runs: `dst` is the workspace, partial s a compact Ga x Gb matrix of its own - ldo is Gb then)
template <int XT, int YT>
__device__ __forceinline__ void gram_write(const v4d_t (&acc)[XT][YT], double *__restrict__ dst, bool mirror, int Ga, int Gb, int64_t ldo, int ci, int cj,
                                           int lrow, int lcol)
{
#pragma unroll
    for (int x = 0; x < XT; ++x)
#pragma unroll
        for (int y = 0; y < YT; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = ci + x * 16 + lrow + 4 * r, j = cj + y * 16 + lcol;
                if (i < Ga && j < Gb) {
                    dst[(int64_t)i * ldo + j] = acc[x][y][r];
                    if (mirror) dst[(int64_t)j * ldo + i] = acc[x][y][r];
                }
            }
}

// the (split, tile) a workgroup owns.  XCD-aware: workgroup b runs on XCD b % 8 (observed; speed only) -> an XCD owns a contiguous range of
// (split, tile): tiles next to each other share a row panel of A in one L2
template <bool SYM> __device__ __forceinline__ bool gram_tile_of(int nta, int ntb, int ksplit, int &s, int &it, int &jt)
{
    const int ntile = SYM ? nta * (nta + 1) / 2 : nta * ntb;
    const int total = ntile * ksplit, per = (total + 7) / 8;
    const int q = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    if (q >= total) return false;
    s = q / ntile;
    const int p = q - s * ntile;
    if (SYM) {
        it = 0;
        int rem = p;
        while (rem >= nta - it) { rem -= nta - it; ++it; }
        jt = it + rem;
    } else {
        it = p / ntb;
        jt = p - it * ntb;
    }
    return true;
}

// out (Ga x Gb) = (A - ma)^T (B - mb); A (C, lda) of TA, B (C, ldb) of TB.  SYM: B is A (Gb == Ga), only tile pairs it <= jt.
// TN: columns of the output tile (128: waves 2 x 2, each 64 x 64; 64: waves 4 x 1, each 32 x 64 - the thin blocks of the
// subspace iteration, where a 128-wide tile would multiply mostly padding).
template <typename TA, typename TB, bool SYM, int TN>
__global__ __launch_bounds__(GM_THREADS, 2) void k_gram(const TA *__restrict__ A, const TB *__restrict__ B, const double *__restrict__ ma,
                                                         const double *__restrict__ mb, double *__restrict__ out, int C, int Ga, int Gb,
                                                         int64_t lda, int64_t ldb, int64_t ldo, int nta, int ntb, int ksplit, int cells_per_split,
                                                         int64_t part_stride)
{
    static_assert(TN == 128 || TN == 64, "tile widths");
    constexpr int XT = TN == 128 ? 4 : 2, YT = 4;                     // 16 x 16 MFMA tiles per wave along i and j
    constexpr int LDB = TN + 16;                                      // LDS pitch of the B slab
    constexpr int PB = TN / 2, RB = GM_THREADS / PB, UB = GM_KS / RB; // B staging: column pairs per row, rows per pass, passes
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *As = reinterpret_cast<double *>(smem);                    // [2][KS][GM_LD]
    double *Bs = As + 2 * GM_KS * GM_LD;                              // [2][KS][LDB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int s, it, jt;
    if (!gram_tile_of<SYM>(nta, ntb, ksplit, s, it, jt)) return;
    const int i0 = it * GM_T, j0 = jt * TN;
    const int c_begin = s * cells_per_split, c_end = min(C, c_begin + cells_per_split);
    // staging roles.  A slab: column pair cp of the tile (genes 2 cp, 2 cp + 1), rows r0 + 4 u; B slab: pair cpb, rows r0b + RB u
    const int cp = tid & 63, r0 = tid >> 6, cpb = tid % PB, r0b = tid / PB;
    const int ga = i0 + 2 * cp, gb = j0 + 2 * cpb;
    const bool a_ok0 = ga < Ga, a_ok1 = ga + 1 < Ga, b_ok0 = gb < Gb, b_ok1 = gb + 1 < Gb;
    const double ma0 = (ma && a_ok0) ? ma[ga] : 0.0, ma1 = (ma && a_ok1) ? ma[ga + 1] : 0.0;
    const double mb0 = (mb && b_ok0) ? mb[gb] : 0.0, mb1 = (mb && b_ok1) ? mb[gb + 1] : 0.0;
    // rows are padded to an even pitch >= the column count: a pair starting at an even column below the pitch lies inside its row
    const int ca = ga < lda ? ga : 0, cb = gb < ldb ? gb : 0;
    // fetch() only ISSUES the loads of a slab (raw values into registers); the centring, the masking and the LDS writes wait in
    // stash(), after the slab in hand has been multiplied - a subtract right behind its load would put the wait for the load in
    // front of the matrix instructions it is meant to hide behind
    double ra[4][2], rb[UB][2];

    auto fetch = [&](int c0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + r0 + 4 * u;
            load2<TA>(A + (int64_t)min(c, c_end - 1) * lda + ca, ra[u][0], ra[u][1]);
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {                                // (a diagonal tile of the Gram matrix stages the same slab twice:
            const int c = c0 + r0b + RB * u;                          //  one tile in nta, and no branch around a load)
            load2<TB>(B + (int64_t)min(c, c_end - 1) * ldb + cb, rb[u][0], rb[u][1]);
        }
    };
    auto stash = [&](int buf, int c0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + 4 * u;
            const bool in = c0 + r < c_end;
            const double a0 = (in && a_ok0) ? ra[u][0] - ma0 : 0.0, a1 = (in && a_ok1) ? ra[u][1] - ma1 : 0.0;
            *reinterpret_cast<double2 *>(As + (buf * GM_KS + r) * GM_LD + 2 * cp) = double2{a0, a1};
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int r = r0b + RB * u;
            const bool in = c0 + r < c_end;
            const double b0 = (in && b_ok0) ? rb[u][0] - mb0 : 0.0, b1 = (in && b_ok1) ? rb[u][1] - mb1 : 0.0;
            *reinterpret_cast<double2 *>(Bs + (buf * GM_KS + r) * LDB + 2 * cpb) = double2{b0, b1};
        }
    };

    const int wm = TN == 128 ? wave >> 1 : wave, wn = TN == 128 ? wave & 1 : 0;
    const int wi = wm * XT * 16, wj = wn * YT * 16;                   // the wave's corner inside the tile
    v4d_t acc[XT][YT];
#pragma unroll
    for (int x = 0; x < XT; ++x)
#pragma unroll
        for (int y = 0; y < YT; ++y) acc[x][y] = v4d_t{0.0, 0.0, 0.0, 0.0};
    const int lrow = lane >> 4, lcol = lane & 15;

    int buf = 0;
    if (c_begin < c_end) {
        fetch(c_begin);
        stash(0, c_begin);
    }
    __syncthreads();
    for (int c0 = c_begin; c0 < c_end; c0 += GM_KS) {
        const bool more = c0 + GM_KS < c_end;
        if (more) fetch(c0 + GM_KS);                                  // next slab in flight while this one is multiplied
        __builtin_amdgcn_sched_barrier(0);
        const double *as = As + buf * GM_KS * GM_LD, *bs = Bs + buf * GM_KS * LDB;
#pragma unroll
        for (int kk = 0; kk < GM_KS / 4; ++kk) {
            double a[XT], b[YT];
#pragma unroll
            for (int x = 0; x < XT; ++x) a[x] = as[(kk * 4 + lrow) * GM_LD + wi + x * 16 + lcol];
#pragma unroll
            for (int y = 0; y < YT; ++y) b[y] = bs[(kk * 4 + lrow) * LDB + wj + y * 16 + lcol];
#pragma unroll
            for (int x = 0; x < XT; ++x)
#pragma unroll
                for (int y = 0; y < YT; ++y) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x], b[y], acc[x][y], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) stash(buf ^ 1, c0 + GM_KS);
        __syncthreads();
        buf ^= 1;
    }
    gram_write<XT, YT>(acc, out + (ksplit > 1 ? (int64_t)s * part_stride : 0), SYM && ksplit == 1 && it != jt, Ga, Gb, ldo, i0 + wi, j0 + wj, lrow, lcol);
}

// The same product with the slabs DMA'd from global memory straight into LDS (global_load_lds_dwordx4), the form the linear all-pairs
// kernel below arrived at first: no staging registers, no ds_write in the wave's instruction stream, nothing between two slabs' matrix
// instructions but the wait.  One LDS-DMA instruction lays its 64 lanes' 16-byte pieces down back to back = 1 KiB: one slab row of 128 f64
// genes, or two rows of 512 bytes (128 f32 genes, 64 f64 columns of a thin block) - lanes 0..31 take cell b, lanes 32..63 cell b + 8, so that
// the four cells of one fragment read (4 kk + (lane >> 4)) always sit in four DIFFERENT instruction blocks, and the blocks are laid out with
// a pitch of 1 KiB + 128 / 64 bytes: the four 16-lane groups of a fragment read start 32 / 16 banks apart.  What the DMA cannot do on the way
// is arithmetic, so the RAW values land in LDS (in the storage type) and the centring moves to the fragment read: one f64 subtract (and a
// convert for f32 storage) per fragment element, 8 per 16 matrix instructions, with the 8 means of the wave's rows and columns in registers.
// Cells past the end of the split (the last slab of the last split) are clamped copies of its last cell; their fragment elements are
// zeroed after the centring.  Tile columns past Ga / Gb hold whatever lies there (inside the row, or column 0 beyond the pitch): a column of
// the operands only reaches its own row / column of the output, which is never written.
template <typename T, int W> struct GramSlab {                       // a slab of GM_KS cells x W columns of T in LDS-DMA order
    static constexpr int ROWB = W * (int)sizeof(T);                   // bytes of a slab row
    static constexpr int RPI = 1024 / ROWB;                           // rows per DMA instruction: 1 or 2
    static_assert(RPI == 1 || RPI == 2, "slab rows of 1 KiB or 512 bytes");
    static constexpr int BLK = 1024 + (RPI == 1 ? 128 : 64);          // pitch of the instruction blocks
    static constexpr int NBLK = GM_KS / RPI;                          // instructions per slab
    static constexpr int BYTES = NBLK * BLK;
    static constexpr int PIECE = 16 / (int)sizeof(T);                 // elements per lane of an instruction
    __device__ static __forceinline__ int cell_of(int blk, int lane) { return RPI == 1 ? blk : blk + 8 * (lane >> 5); }
    __device__ static __forceinline__ int col_of(int lane) { return (RPI == 1 ? lane : (lane & 31)) * PIECE; }
    __device__ static __forceinline__ int at(int cell, int col) { return (RPI == 1 ? cell * BLK : (cell & 7) * BLK + (cell >> 3) * 512) + col * (int)sizeof(T); }
};

template <typename TA, typename TB, bool SYM, int TN>
__global__ __launch_bounds__(GM_THREADS, 2) void k_gram_dma(const TA *__restrict__ A, const TB *__restrict__ B, const double *__restrict__ ma,
                                                             const double *__restrict__ mb, double *__restrict__ out, int C, int Ga, int Gb,
                                                             int64_t lda, int64_t ldb, int64_t ldo, int nta, int ntb, int ksplit, int cells_per_split,
                                                             int64_t part_stride)
{
    static_assert(TN == 128 || TN == 64, "tile widths");
    constexpr int XT = TN == 128 ? 4 : 2, YT = 4;
    using SA = GramSlab<TA, GM_T>;
    using SB = GramSlab<TB, TN>;
    constexpr int BUF = SA::BYTES + SB::BYTES;
    // two arrays, not two halves of one: the compiler orders every LDS read behind the LDS-DMA writes it cannot tell apart from it
    __shared__ __attribute__((aligned(16))) char buf0[BUF];
    __shared__ __attribute__((aligned(16))) char buf1[BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int s, it, jt;
    if (!gram_tile_of<SYM>(nta, ntb, ksplit, s, it, jt)) return;
    const int i0 = it * GM_T, j0 = jt * TN;
    const int c_begin = s * cells_per_split, c_end = min(C, c_begin + cells_per_split);
    const int wm = TN == 128 ? wave >> 1 : wave, wn = TN == 128 ? wave & 1 : 0;
    const int wi = wm * XT * 16, wj = wn * YT * 16;
    const int lrow = lane >> 4, lcol = lane & 15;
    v4d_t acc[XT][YT];
#pragma unroll
    for (int x = 0; x < XT; ++x)
#pragma unroll
        for (int y = 0; y < YT; ++y) acc[x][y] = v4d_t{0.0, 0.0, 0.0, 0.0};
    if (c_begin < c_end) {
        // the means of this lane's fragment columns (0 without centring and past the last column)
        double mA[XT], mB[YT];
#pragma unroll
        for (int x = 0; x < XT; ++x) { const int g = i0 + wi + x * 16 + lcol; mA[x] = (ma && g < Ga) ? ma[g] : 0.0; }
#pragma unroll
        for (int y = 0; y < YT; ++y) { const int g = j0 + wj + y * 16 + lcol; mB[y] = (mb && g < Gb) ? mb[g] : 0.0; }
        // DMA roles: wave w issues the instruction blocks w, w + 4, ... of both operands
        const int colA = i0 + SA::col_of(lane), colB = j0 + SB::col_of(lane);
        const TA *pA = A + (colA < lda ? colA : 0);
        const TB *pB = B + (colB < ldb ? colB : 0);
        typedef __attribute__((address_space(3))) void lds_void;
        typedef const __attribute__((address_space(1))) void glb_void;
        auto dma = [&](char *base, int c0) {
#pragma unroll
            for (int u = 0; u < SA::NBLK / 4; ++u) {
                const int blk = wave + 4 * u;
                const int c = min(c0 + SA::cell_of(blk, lane), c_end - 1);
                __builtin_amdgcn_global_load_lds((glb_void *)(pA + (int64_t)c * lda), (lds_void *)(base + blk * SA::BLK), 16, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < SB::NBLK / 4; ++u) {
                const int blk = wave + 4 * u;
                const int c = min(c0 + SB::cell_of(blk, lane), c_end - 1);
                __builtin_amdgcn_global_load_lds((glb_void *)(pB + (int64_t)c * ldb), (lds_void *)(base + SA::BYTES + blk * SB::BLK), 16, 0, 0);
            }
        };
        // one slab: the next one streams into `other` while the matrix instructions read `cur`; valid = cells of this slab inside the split
        auto multiply = [&](const char *cur, auto tail, int valid) {
            const char *as = cur, *bs = cur + SA::BYTES;
#pragma unroll
            for (int kk = 0; kk < GM_KS / 4; ++kk) {
                const int cell = kk * 4 + lrow;
                double a[XT], b[YT];
#pragma unroll
                for (int x = 0; x < XT; ++x) a[x] = (double)*reinterpret_cast<const TA *>(as + SA::at(cell, wi + x * 16 + lcol)) - mA[x];
#pragma unroll
                for (int y = 0; y < YT; ++y) b[y] = (double)*reinterpret_cast<const TB *>(bs + SB::at(cell, wj + y * 16 + lcol)) - mB[y];
                if constexpr (decltype(tail)::value) {                // the slab a split ends inside: cells past its end count as zeros
                    const bool in = cell < valid;
#pragma unroll
                    for (int x = 0; x < XT; ++x) a[x] = in ? a[x] : 0.0;
#pragma unroll
                    for (int y = 0; y < YT; ++y) b[y] = in ? b[y] : 0.0;
                }
#pragma unroll
                for (int x = 0; x < XT; ++x)
#pragma unroll
                    for (int y = 0; y < YT; ++y) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x], b[y], acc[x][y], 0, 0, 0);
            }
        };
        // one whole slab: the next one streams into `other` while the matrix instructions read `cur`
        auto slab = [&](const char *cur, char *other, int c0) {
            if (c0 + GM_KS < c_end) dma(other, c0 + GM_KS);           // every wave is past the barrier that ended the reads of that buffer
            __builtin_amdgcn_sched_barrier(0);
            multiply(cur, std::false_type{}, GM_KS);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's pieces of the next slab have landed in LDS
            __syncthreads();
        };
        dma(buf0, c_begin);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int c0 = c_begin;
        bool second = false;                                          // which buffer holds the slab at c0
        for (; c0 + 2 * GM_KS <= c_end; c0 += 2 * GM_KS) {
            slab(buf0, buf1, c0);
            slab(buf1, buf0, c0 + GM_KS);
        }
        if (c0 + GM_KS <= c_end) { slab(buf0, buf1, c0); c0 += GM_KS; second = true; }
        if (c0 < c_end) multiply(second ? buf1 : buf0, std::true_type{}, c_end - c0);
    }
    gram_write<XT, YT>(acc, out + (ksplit > 1 ? (int64_t)s * part_stride : 0), SYM && ksplit == 1 && it != jt, Ga, Gb, ldo, i0 + wi, j0 + wj, lrow, lcol);
}

// out[i][j] = sum over the splits, in split order; SYM: the lower triangle's tiles are read from their mirror images
template <bool SYM>
__global__ __launch_bounds__(256) void k_gram_reduce(const double *__restrict__ part, double *__restrict__ out, int Ga, int Gb, int64_t ldo, int ksplit,
                                                      int64_t part_stride)
{
    const int64_t n = (int64_t)Ga * Gb;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(t / Gb), j = (int)(t - (int64_t)i * Gb);
        const bool flip = SYM && (i / GM_T) > (j / GM_T);
        const int64_t src = flip ? (int64_t)j * Gb + i : (int64_t)i * Gb + j;            // partials are compact (row pitch Gb)
        double v = 0.0;
        for (int s = 0; s < ksplit; ++s) v += part[(int64_t)s * part_stride + src];
        out[(int64_t)i * ldo + j] = v;
    }
}

// column means of a cells-major matrix, fp64, fixed order: block b sums rows b, b + nb, ...; the partials are folded in block order
template <typename T>
__global__ __launch_bounds__(256) void k_col_sums_partial(const T *__restrict__ X, double *__restrict__ part, int C, int G, int64_t ld, int nb)
{
    const int g = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (g >= G) return;
    double s = 0.0;
    for (int c = b; c < C; c += nb) s += (double)X[(int64_t)c * ld + g];
    part[(int64_t)b * G + g] = s;
}
__global__ __launch_bounds__(256) void k_col_means_fold(const double *__restrict__ part, double *__restrict__ mean, int C, int G, int nb)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    double s = 0.0;
    for (int b = 0; b < nb; ++b) s += part[(int64_t)b * G + g];
    mean[g] = s / (double)C;
}

constexpr int GM_MEAN_BLOCKS = 64;

static int gram_ksplit(int64_t ntile, int64_t C, int64_t Ga, int64_t Gb, int cus)
{
    // Two workgroups are resident per CU and every workgroup of a launch costs the same, so a launch runs in ROUNDS of 2 x CUs
    // workgroups and its last round should be full: among the splits that leave at least 1024 cells per workgroup, keep the
    // partial tiles within 2 GiB and give at most ~8 rounds, take the largest whose last round is at least 93 % full (else the
    // fullest).  300 tiles (3000 genes) on 256 CUs: 5 splits = 2.93 rounds, where 7 splits were 4.1 rounds = 5 rounds of time.
    const int64_t slots = 2LL * (cus > 0 ? cus : 256);
    if (ntile >= 8 * slots) return 1;
    int64_t kmax = (8 * slots + ntile - 1) / ntile;
    const int64_t by_cells = C / 1024 > 0 ? C / 1024 : 1;
    if (kmax > by_cells) kmax = by_cells;
    const int64_t by_mem = (int64_t)(2LL << 30) / (Ga * Gb * 8 > 0 ? Ga * Gb * 8 : 1);
    if (kmax > by_mem) kmax = by_mem;
    if (kmax < 1) kmax = 1;
    auto fill = [&](int64_t ks) {
        const double rounds = (double)(ntile * ks) / (double)slots;
        return rounds / (double)(int64_t)(rounds + 0.999999);
    };
    int64_t best = 1;
    double best_fill = fill(1);
    for (int64_t ks = 2; ks <= kmax; ++ks) {
        const double f = fill(ks);
        if (f >= 0.93) { best = ks; if (f > best_fill) best_fill = f; }           // the largest split with a full last round
        else if (best_fill < 0.93 && f > best_fill) { best = ks; best_fill = f; }  // none so far: the fullest
    }
    return (int)best;
}

static inline int gram_tile_cols(int64_t Gb, bool sym) { return (!sym && Gb <= 64) ? 64 : 128; }

template <typename TA, typename TB, bool SYM, int TN>
static int launch_gram_t(const void *A, const void *B, const double *ma, const double *mb, double *out, void *ws, int64_t C, int64_t Ga, int64_t Gb,
                         int64_t lda, int64_t ldb, int64_t ldo, hipStream_t st)
{
    DevInfo dev;
    int rc = device_info(&dev);
    if (rc) return rc;
    const int64_t nta = (Ga + GM_T - 1) / GM_T, ntb = (Gb + TN - 1) / TN;
    const int64_t ntile = SYM ? nta * (nta + 1) / 2 : nta * ntb;
    const int ksplit = gram_ksplit(ntile, C, Ga, Gb, dev.cus);
    int64_t cps = (C + ksplit - 1) / ksplit;
    cps = (cps + GM_KS - 1) / GM_KS * GM_KS;
    const int64_t total = ntile * ksplit, blocks = (total + 7) / 8 * 8;
    if (blocks >= (1LL << 31)) return fail(VCY_ERR_INVALID, "%s: grid too large", "gram");
    if (ksplit > 1 && !ws) return fail(VCY_ERR_INVALID, "%s: workspace missing (vcy_gram_workspace_bytes)", "gram");
    const int64_t part_stride = Ga * Gb;
    if (env_int("VCY_GRAM_DMA", 1) != 0) {                            // (0: the register-staged form, for A/B)
        hipLaunchKernelGGL((k_gram_dma<TA, TB, SYM, TN>), dim3((unsigned)blocks), dim3(GM_THREADS), 0, st, (const TA *)A, (const TB *)B, ma, mb,
                           ksplit > 1 ? (double *)ws : out, (int)C, (int)Ga, (int)Gb, lda, ldb, ksplit > 1 ? Gb : ldo, (int)nta, (int)ntb, ksplit, (int)cps,
                           part_stride);
    } else {
        const size_t lds = (size_t)2 * GM_KS * (GM_LD + TN + 16) * sizeof(double);
        auto kern = k_gram<TA, TB, SYM, TN>;
        rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
        if (rc) return rc;
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(GM_THREADS), lds, st, (const TA *)A, (const TB *)B, ma, mb, ksplit > 1 ? (double *)ws : out, (int)C,
                           (int)Ga, (int)Gb, lda, ldb, ksplit > 1 ? Gb : ldo, (int)nta, (int)ntb, ksplit, (int)cps, part_stride);
    }
    VCY_LAUNCH_CHECK();
    if (ksplit > 1) {
        const int64_t n = Ga * Gb;
        const int rb = (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
        hipLaunchKernelGGL(k_gram_reduce<SYM>, dim3(rb), dim3(256), 0, st, (const double *)ws, out, (int)Ga, (int)Gb, ldo, ksplit, part_stride);
        VCY_LAUNCH_CHECK();
    }       // (ksplit == 1: the kernel wrote `out` itself and mirrored its off-diagonal tiles)
    return VCY_OK;
}

template <typename TA, typename TB, bool SYM>
static int launch_gram(const void *A, const void *B, const double *ma, const double *mb, double *out, void *ws, int64_t C, int64_t Ga, int64_t Gb,
                       int64_t lda, int64_t ldb, int64_t ldo, hipStream_t st)
{
    if constexpr (!SYM) {
        if (gram_tile_cols(Gb, false) == 64) return launch_gram_t<TA, TB, false, 64>(A, B, ma, mb, out, ws, C, Ga, Gb, lda, ldb, ldo, st);
    }
    return launch_gram_t<TA, TB, SYM, 128>(A, B, ma, mb, out, ws, C, Ga, Gb, lda, ldb, ldo, st);
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The linear all-pairs variant, speedboosted._colDeltaCor (speedboosted.pyx:13-87; wrapper estimation.py:11-33): without a transform the
// correlation of A = e_i - e_c with b = d_c over the genes is separable,
//     sum A = Se_i - Se_c,    sum A^2 = See_i + See_c - 2 (E E^T)[c][i],    sum A b = (D E^T)[c][i] - sum_g e_c[g] d_c[g],
// i.e. two products that contract over the GENES, the contiguous dimension of both operands ("NT"): the one place on the path that is a
// genuine dense block contraction, done here on the f64 matrix cores with the moment algebra, the noise-floor NaN rule and the
// `rm[c][i] (+)= r` store fused into the epilogue - no (rows x C) temporaries, no library GEMM.
// A workgroup of 4 waves owns a 128 (cells c) x 64 (cells i) tile of rm and both products of it: wave (wm, wn) owns 64 x 32 = 4 x 2
// MFMA tiles of each product = 64 f64 accumulators per lane.  Slabs of 16 genes of the 128 + 128 + 64 rows are staged into LDS
// [row][gene] with an odd row pitch (17 doubles): an A- or B-fragment of v_mfma_f64_16x16x4_f64 is row (lane & 15), gene (lane >> 4) of a
// 16 x 4 block, and the 16 rows of a quarter-wave start in 16 different bank pairs.  The expansion of sum A^2 cancels where e_i is
// close to e_c - which is why everything is f64 whatever the storage type.
// Two forms.  k_cdc_full_linear_dma (below; the one that runs whenever the row pitch holds whole 128-byte slab rows, i.e. always with the
// layout's 64-element padding): slabs DMA'd from global memory straight into LDS.  Measured at 10 000 cells x 20 000 genes
// (tools/bench_full.py, profiles/r05_full_linear.txt): f64 storage 120.4 ms = 66.4 Tflop/s = 0.85 of the f64 matrix peak (0.87 at the 2.33 GHz
// of the launch), f32 storage 118.0 ms; the same algebra as two library GEMMs + eager elementwise passes (round 4's route) 122.2 ms; the
// element-wise kernel 427 ms.  k_cdc_full_linear (this one; the fallback for other pitches, VCY_NT_DMA=0 for A/B): slabs staged through
// registers, 146-150 ms.  How the first became the second: 155 ms with masked LDS writes after the slab's matrix instructions; 149.7 without
// the masks (nothing to mask: the contraction runs over zero-padded genes); 146 with the LDS writes issued inside the slab's matrix
// instructions.  Cost probes on that form (results wrong, cost right): no loads and no LDS writes after the first slab 115.7 ms - the
// matrix / fragment-read / barrier skeleton is worth 0.88 of the peak -, LDS writes of stale registers 133.9, i.e. the ds_write instructions
// cost 16 % of the launch and the wait for their loads 10 %; tile width (64 / 128 cells i), LDS pitch (17 / 18 doubles), 8- or 16-byte LDS
// accesses, issuing the loads a slab earlier: no change.  So the staged slab had to reach LDS without passing through the wave:
// global_load_lds_dwordx4.  VCY_NT_N / VCY_NT_KS / VCY_NT_PAD rebuild variants of the register-staged form.
#ifndef VCY_NT_N
#define VCY_NT_N 64
#endif
#ifndef VCY_NT_KS
#define VCY_NT_KS 16
#endif
#ifndef VCY_NT_PAD
#define VCY_NT_PAD 2
#endif
static_assert(VCY_NT_PAD % 2 == 0 && VCY_NT_KS % 8 == 0, "16-byte LDS accesses");
static_assert(2 * (2 * 128 + VCY_NT_N) * (VCY_NT_KS + VCY_NT_PAD) * 8 <= 160 * 1024, "the double-buffered slabs must fit the LDS of a CU");
constexpr int NT_M = 128, NT_N = VCY_NT_N, NT_KS = VCY_NT_KS, NT_LD = NT_KS + VCY_NT_PAD;
constexpr int NT_YT = NT_N / 32;       // 16 x 16 tiles per wave along i (waves 2 x 2)

// per cell: Se = sum e, See = sum e^2, sb = sum d, sbb = sum d^2, sed = sum e d   (f64, one workgroup per cell, fixed order)
template <typename T> __global__ __launch_bounds__(256) void k_cell_linear_sums(const T *__restrict__ e, const T *__restrict__ d, double *__restrict__ sums,
                                                                                  int G, int64_t ld, int64_t cell0, int C_out)
{
    __shared__ double red[8];
    const int64_t c = blockIdx.x;
    const T *er = e + c * ld;
    const bool own = c >= cell0 && c < cell0 + C_out;                 // d holds the same rows as e; only the launch's own cells need its sums
    const T *dr = d + c * ld;
    double se = 0.0, see = 0.0, sb = 0.0, sbb = 0.0, sed = 0.0;
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        const double x = (double)er[g];
        se += x; see = fma(x, x, see);
        if (own) { const double y = (double)dr[g]; sb += y; sbb = fma(y, y, sbb); sed = fma(x, y, sed); }
    }
    se = block_sum(se, red); see = block_sum(see, red);
    sb = block_sum(sb, red); sbb = block_sum(sbb, red); sed = block_sum(sed, red);
    if (threadIdx.x == 0) { double *o = sums + 5 * c; o[0] = se; o[1] = see; o[2] = sb; o[3] = sbb; o[4] = sed; }
}

// the epilogue of both forms of the kernel: D[row = (lane >> 4) + 4 reg][col = lane & 15] of every 16 x 16 tile -> Pearson's r of the pair (c, i).
// The expansion of sum A^2 (and of sum A b) loses what the terms that cancel carry in their last bits: the absolute error of va is a few
// ulps x sqrt(G) of See_i + See_c, so the RELATIVE error of r grows like (See_i + See_c) / va - nearly identical cells, the population a kNN
// graph is made of.  The reference centres A = e_i - e_c before it squares (speedboosted.pyx:29-78) and has no such loss.  Pairs whose
// variance is below NT_TAU of the terms it was expanded from are therefore NOT written here: their bit is set in `flags` (one uint16 per row
// and 16 columns = one ballot quarter, written whole by the lane of column 0: no atomics, no clearing pass) and k_cdc_linear_repair
// re-evaluates exactly those pairs in the reference's own centred two-pass form.  NaN is then only ever the reference's 0 * inf:
// i == c here, exact duplicates and a constant d_c in the repair pass.
constexpr double NT_TAU = 1.0 / 1024.0;    // error of r from the expansion <= ~1e-13 (See_i + See_c) / va: below 1e-10 above this ratio

template <typename OT>
__device__ __forceinline__ void nt_epilogue(const v4d_t (&accE)[4][NT_YT], const v4d_t (&accD)[4][NT_YT], const double *__restrict__ sums, OT *__restrict__ rm,
                                            unsigned short *__restrict__ flags, int64_t flag_pitch, int C, int G, int64_t cell0, int C_out, int64_t ld_rm,
                                            int accumulate, int c0, int i0, int wi, int wj, int lrow, int lcol)
{
    const double n = (double)G;
#pragma unroll
    for (int y = 0; y < NT_YT; ++y) {
        const int i = i0 + wj + y * 16 + lcol;
        const bool iok = i < C;
        const double Se_i = iok ? sums[5 * (int64_t)i] : 0.0, See_i = iok ? sums[5 * (int64_t)i + 1] : 0.0;
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cl = c0 + wi + x * 16 + lrow + 4 * r;
                const bool ok = iok && cl < C_out;
                bool redo = false;
                if (ok) {
                    const double *sc = sums + 5 * (cell0 + cl);
                    const double Se_c = sc[0], See_c = sc[1], sb = sc[2], sbb = sc[3], sed = sc[4];
                    const double sA = Se_i - Se_c;
                    const double sAA = See_i + See_c - 2.0 * accE[x][y][r];
                    const double sAb = accD[x][y][r] - sed;
                    const double cov = sAb - sA * sb / n, va = sAA - sA * sA / n, vb = sbb - sb * sb / n;
                    const bool self = (int64_t)i == cell0 + cl;
                    // (!(a >= b): a NaN or infinite moment goes to the repair pass as well, which reproduces the reference's arithmetic on it)
                    redo = !self && (!(va >= NT_TAU * (See_i + See_c)) || !(vb >= NT_TAU * sbb));
                    if (!redo) {
                        const double v = self ? __builtin_nan("") : cov / sqrt(va * vb);
                        OT *o = rm + (int64_t)cl * ld_rm + i;
                        *o = accumulate ? (OT)((double)*o + v) : (OT)v;
                    }
                }
                const unsigned long long m = __ballot(redo);                 // bits 16 q .. 16 q + 15: row lrow = q of this quad of rows, columns lcol
                if (lcol == 0 && cl < C_out)
                    flags[(int64_t)cl * flag_pitch + ((i0 + wj + y * 16) >> 4)] = (unsigned short)(m >> (16 * lrow));
            }
    }
}

// The repair pass of the linear all-pairs variant: every flagged pair again, as the reference evaluates it (speedboosted.pyx:29-78) - A = e_i - e_c
// formed element by element, centred on its mean, then squared; b = d_c likewise - one wave per pair, f64, two passes over the three rows.
// A workgroup owns one cell c: its waves walk the row's flag words (16 columns each) and take the set bits in turn.  A launch without
// flagged pairs reads C_out x C / 8 bytes of flags and ends.
template <typename T>
__global__ __launch_bounds__(256) void k_cdc_linear_repair(const T *__restrict__ e, const T *__restrict__ d, const unsigned short *__restrict__ flags,
                                                           int64_t flag_pitch, T *__restrict__ rm, int C, int G, int64_t ld, int64_t cell0, int64_t ld_rm,
                                                           int accumulate)
{
    const int cl = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const unsigned short *fr = flags + (int64_t)cl * flag_pitch;
    const int nwords = (C + 15) >> 4;
    const T *ec = e + (cell0 + cl) * ld, *dc = d + (cell0 + cl) * ld;
    const double n = (double)G;
    int taken = 0;                                                     // flagged pairs of this row met so far (the same count in every wave)
    for (int w0 = 0; w0 < nwords; w0 += 64) {
        const unsigned mine = (w0 + lane < nwords) ? (unsigned)fr[w0 + lane] : 0u;
        unsigned long long any = __ballot(mine != 0);
        while (any) {
            const int wl = __builtin_ctzll(any);
            any &= any - 1;
            unsigned bits = (unsigned)__builtin_amdgcn_readlane((int)mine, wl);
            while (bits) {
                const int b = __builtin_ctz(bits);
                bits &= bits - 1;
                if ((taken++ % nw) != wave) continue;
                const int i = ((w0 + wl) << 4) + b;
                const T *ei = e + (int64_t)i * ld;
                double sA = 0.0, sb = 0.0;
                for (int g = lane; g < G; g += 64) { sA += (double)ei[g] - (double)ec[g]; sb += (double)dc[g]; }
                const double muA = wave_sum(sA) / n, mub = wave_sum(sb) / n;
                double ssA = 0.0, ssb = 0.0, sab = 0.0;
                for (int g = lane; g < G; g += 64) {
                    const double a = ((double)ei[g] - (double)ec[g]) - muA, bb = (double)dc[g] - mub;
                    ssA = fma(a, a, ssA); ssb = fma(bb, bb, ssb); sab = fma(a, bb, sab);
                }
                ssA = wave_sum(ssA); ssb = wave_sum(ssb); sab = wave_sum(sab);
                // sum_j (A_mA[j] / sqrt(ssA)) (b_mb[j] / sqrt(ssb)): zero variance gives 0 * inf = NaN, as the reference's products do
                const double v = (sab * (1.0 / sqrt(ssA))) * (1.0 / sqrt(ssb));
                if (lane == 0) {
                    T *o = rm + (int64_t)cl * ld_rm + i;
                    *o = accumulate ? (T)((double)*o + v) : (T)v;
                }
            }
        }
    }
}

template <typename T, typename OT>
__global__ __launch_bounds__(GM_THREADS, NT_N == 128 ? 1 : 2) void k_cdc_full_linear(const T *__restrict__ e, const T *__restrict__ d, const double *__restrict__ sums,
                                                                    OT *__restrict__ rm, unsigned short *__restrict__ flags, int64_t flag_pitch, int C, int G, int64_t ld,
                                                                    int64_t cell0, int C_out, int64_t ld_rm, int accumulate, int ntn)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *Es = reinterpret_cast<double *>(smem);                    // [2][NT_M][NT_LD]   e of the tile's cells c
    double *Ds = Es + 2 * NT_M * NT_LD;                               // [2][NT_M][NT_LD]   d of the tile's cells c
    double *Bs = Ds + 2 * NT_M * NT_LD;                               // [2][NT_N][NT_LD]   e of the tile's cells i
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // workgroup b runs on XCD b % 8 (observed; speed only): an XCD owns a contiguous range of tiles, walked along i inside a row panel
    const int total = (int)gridDim.x, per = (total + 7) / 8;
    const int q = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    const int mt = q / ntn, nt = q - mt * ntn;
    if (mt * NT_M >= C_out) return;
    const int c0 = mt * NT_M, i0 = nt * NT_N;
    // staging: 8 gene pairs per row (16 genes: one 128-byte line of an f64 row), 32 rows per pass
    constexpr int PR = NT_KS / 2, RP = GM_THREADS / PR, UA = NT_M / RP, UB = NT_N / RP;
    const int cp = tid % PR, r0 = tid / PR;
    double ra[UA][2], rd[UA][2], rb[UB][2];
    auto fetch = [&](int g0) {
        const int g = g0 + 2 * cp;
        const int gc = g < ld ? g : 0;                                // rows are padded to an even pitch: a pair below the pitch lies inside its row
#pragma unroll
        for (int u = 0; u < UA; ++u) {
            const int64_t row = cell0 + min(c0 + r0 + RP * u, C_out - 1);
            load2<T>(e + row * ld + gc, ra[u][0], ra[u][1]);
            load2<T>(d + row * ld + gc, rd[u][0], rd[u][1]);
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int64_t row = min(i0 + r0 + RP * u, C - 1);
            load2<T>(e + row * ld + gc, rb[u][0], rb[u][1]);
        }
    };
    // no masks on the way into LDS: the contraction runs over the genes, and the rows of e and d are zero beyond G up to their pitch (the
    // layout's invariant, velocyto_hip.h) - a slab that ends past G adds zeros; rows past the last cell are clamped copies whose outputs
    // the epilogue never writes
    auto stash = [&](int buf, int g0) {
        (void)g0;
#pragma unroll
        for (int u = 0; u < UA; ++u) {
            const int r = r0 + RP * u;
            *reinterpret_cast<double2 *>(Es + (buf * NT_M + r) * NT_LD + 2 * cp) = double2{ra[u][0], ra[u][1]};
            *reinterpret_cast<double2 *>(Ds + (buf * NT_M + r) * NT_LD + 2 * cp) = double2{rd[u][0], rd[u][1]};
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int r = r0 + RP * u;
            *reinterpret_cast<double2 *>(Bs + (buf * NT_N + r) * NT_LD + 2 * cp) = double2{rb[u][0], rb[u][1]};
        }
    };
    const int wm = wave >> 1, wn = wave & 1;
    const int wi = wm * 64, wj = wn * (NT_N / 2);                     // the wave's corner inside the tile
    v4d_t accE[4][NT_YT], accD[4][NT_YT];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < NT_YT; ++y) { accE[x][y] = v4d_t{0.0, 0.0, 0.0, 0.0}; accD[x][y] = v4d_t{0.0, 0.0, 0.0, 0.0}; }
    const int lrow = lane >> 4, lcol = lane & 15;
    int buf = 0;
    fetch(0);
    stash(0, 0);
    if (NT_KS < G) fetch(NT_KS);                                      // the loads of a slab are issued as soon as the staging registers are free:
    __syncthreads();                                                  // a whole slab of matrix instructions ahead of the LDS writes that wait for them
    for (int g0 = 0; g0 < G; g0 += NT_KS) {
        const bool more = g0 + NT_KS < G;
        const double *es = Es + buf * NT_M * NT_LD, *ds = Ds + buf * NT_M * NT_LD, *bs = Bs + buf * NT_N * NT_LD;
        // one 16-byte LDS read feeds TWO matrix instructions: the contraction index may be visited in any order as long as both operands agree, so
        // lane group q = lane >> 4 takes genes 8 kk2 + 2 q and 8 kk2 + 2 q + 1 of the slab - adjacent doubles - for the steps 2 kk2 and 2 kk2 + 1
#pragma unroll
        for (int kk2 = 0; kk2 < NT_KS / 8; ++kk2) {
            double2 a[4], b2[4], b[NT_YT];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                a[x] = *reinterpret_cast<const double2 *>(es + (wi + x * 16 + lcol) * NT_LD + kk2 * 8 + 2 * lrow);
                b2[x] = *reinterpret_cast<const double2 *>(ds + (wi + x * 16 + lcol) * NT_LD + kk2 * 8 + 2 * lrow);
            }
#pragma unroll
            for (int y = 0; y < NT_YT; ++y) b[y] = *reinterpret_cast<const double2 *>(bs + (wj + y * 16 + lcol) * NT_LD + kk2 * 8 + 2 * lrow);
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < NT_YT; ++y) {
                    accE[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x].x, b[y].x, accE[x][y], 0, 0, 0);
                    accD[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(b2[x].x, b[y].x, accD[x][y], 0, 0, 0);
                }
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < NT_YT; ++y) {
                    accE[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x].y, b[y].y, accE[x][y], 0, 0, 0);
                    accD[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(b2[x].y, b[y].y, accD[x][y], 0, 0, 0);
                }
            // the next slab goes into the OTHER buffer while the last matrix instructions of this one are still queued: its loads have had
            // a slab to arrive, and the LDS writes run beside the matrix pipe instead of after it
            if (kk2 == 0) {
                __builtin_amdgcn_sched_barrier(0);
                if (more) stash(buf ^ 1, g0 + NT_KS);
                if (g0 + 2 * NT_KS < G) fetch(g0 + 2 * NT_KS);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        buf ^= 1;
    }
    nt_epilogue<OT>(accE, accD, sums, rm, flags, flag_pitch, C, G, cell0, C_out, ld_rm, accumulate, c0, i0, wi, wj, lrow, lcol);
}


// The same kernel with the slabs DMA'd straight from global memory into LDS (global_load_lds_dwordx4: 16 bytes per lane, no
// staging registers, no ds_write instruction, nothing in the wave's instruction stream between two slabs' matrix instructions but the
// wait).  Measured on the register-staged form: the matrix skeleton alone runs at 0.86-0.88 of the f64 matrix peak, the LDS writes of the
// staged slab cost 16 % of the launch and the wait for their loads 10 % (profiles/r05_full_linear.txt).  An LDS-DMA instruction lays the 64
// lanes' 16-byte pieces down back to back, so a slab row is 8 pieces of 2 genes with NO padding, and the bank spread comes from a swizzle
// instead: piece p of row r holds genes 2 (p ^ (r & 7)), +1.  A fragment read (row = lane & 15 of a 16-row tile, piece 4 kk2 + (lane >> 4))
// then takes eight different pieces per eight rows.  f32 storage: the same 128-byte rows hold 32 genes, a piece is four floats converted
// after the LDS read and feeds four matrix instructions per tile.
constexpr int NTD_ROWB = 128;                                          // bytes of a slab row: 16 genes, f64
template <typename T>
__global__ __launch_bounds__(GM_THREADS, 2) void k_cdc_full_linear_dma(const T *__restrict__ e, const T *__restrict__ d, const double *__restrict__ sums,
                                                                        T *__restrict__ rm, unsigned short *__restrict__ flags, int64_t flag_pitch, int C, int G, int64_t ld,
                                                                        int64_t cell0, int C_out, int64_t ld_rm, int accumulate, int ntn)
{
    static_assert(NT_N == 64, "the DMA form is laid out for 128 x 64 tiles");
    constexpr int PG = 16 / (int)sizeof(T);                           // genes per 16-byte piece: 2 (f64) or 4 (f32: converted after the LDS read)
    constexpr int KSL = 8 * PG;                                       // genes per slab: a slab row is 8 pieces = 128 bytes whatever the type
    constexpr int BUF = (2 * NT_M + NT_N) * NTD_ROWB;                 // one buffer: E rows, D rows, B rows
    // TWO arrays, not two halves of one: the compiler makes every LDS read wait for the LDS-DMA writes it cannot tell apart from the
    // read's address (s_waitcnt vmcnt(0) in front of the slab's first fragment read - the overlap gone); distinct objects it can
    __shared__ __attribute__((aligned(16))) char bufA[BUF];
    __shared__ __attribute__((aligned(16))) char bufB[BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int total = (int)gridDim.x, per = (total + 7) / 8;
    const int q = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    const int mt = q / ntn, nt = q - mt * ntn;
    if (mt * NT_M >= C_out) return;
    const int c0 = mt * NT_M, i0 = nt * NT_N;
    // DMA roles: one instruction = 8 rows x 8 pieces; a wave issues 10 of the 40 a slab needs (4 + 4 for its 32 rows of E and of D, 2 for its
    // 16 rows of B).  Lane l writes piece l & 7 of row l >> 3 of the group; it reads the genes the swizzle puts there.
    const int lr = lane >> 3, lp = lane & 7;
    const T *srcE[4], *srcD[4], *srcB[2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int r = wave * 32 + u * 8 + lr;                         // row of the tile's c block
        const int64_t row = cell0 + min(c0 + r, C_out - 1);
        const int piece = lp ^ (r & 7);
        srcE[u] = e + row * ld + PG * piece;
        srcD[u] = d + row * ld + PG * piece;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int r = wave * 16 + u * 8 + lr;
        const int64_t row = min(i0 + r, C - 1);
        srcB[u] = e + row * ld + PG * (lp ^ (r & 7));
    }
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    auto dma = [&](char *base, int g0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            __builtin_amdgcn_global_load_lds((glb_void *)(srcE[u] + g0), (lds_void *)(base + (wave * 32 + u * 8) * NTD_ROWB), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void *)(srcD[u] + g0), (lds_void *)(base + NT_M * NTD_ROWB + (wave * 32 + u * 8) * NTD_ROWB), 16, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
            __builtin_amdgcn_global_load_lds((glb_void *)(srcB[u] + g0), (lds_void *)(base + 2 * NT_M * NTD_ROWB + (wave * 16 + u * 8) * NTD_ROWB), 16, 0, 0);
    };
    const int wm = wave >> 1, wn = wave & 1;
    const int wi = wm * 64, wj = wn * (NT_N / 2);
    v4d_t accE[4][NT_YT], accD[4][NT_YT];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < NT_YT; ++y) { accE[x][y] = v4d_t{0.0, 0.0, 0.0, 0.0}; accD[x][y] = v4d_t{0.0, 0.0, 0.0, 0.0}; }
    const int lrow = lane >> 4, lcol = lane & 15;
    // one slab: the next one streams into `other` while the matrix instructions read `cur`
    auto slab = [&](const char *cur, char *other, int g0) {
        if (g0 + KSL < G) dma(other, g0 + KSL);                       // every wave is past the barrier that ended the reads of that buffer
        __builtin_amdgcn_sched_barrier(0);
        const char *es = cur, *ds = es + NT_M * NTD_ROWB, *bs = es + 2 * NT_M * NTD_ROWB;
#pragma unroll
        for (int kk2 = 0; kk2 < 2; ++kk2) {                             // 8 pieces per row: lane group q = lane >> 4 reads piece 4 kk2 + q
            struct alignas(16) P { T v[PG]; };
            P a[4], b2[4], b[NT_YT];
            const int piece = ((kk2 * 4 + lrow) ^ (lcol & 7)) * 16;  // (the tiles start at multiples of 16 rows: row & 7 = lcol & 7)
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                a[x] = *reinterpret_cast<const P *>(es + (wi + x * 16 + lcol) * NTD_ROWB + piece);
                b2[x] = *reinterpret_cast<const P *>(ds + (wi + x * 16 + lcol) * NTD_ROWB + piece);
            }
#pragma unroll
            for (int y = 0; y < NT_YT; ++y) b[y] = *reinterpret_cast<const P *>(bs + (wj + y * 16 + lcol) * NTD_ROWB + piece);
#pragma unroll
            for (int j = 0; j < PG; ++j) {                              // one 16-byte read feeds PG matrix instructions per tile
                double aj[4], dj[4], bj[NT_YT];
#pragma unroll
                for (int x = 0; x < 4; ++x) { aj[x] = (double)a[x].v[j]; dj[x] = (double)b2[x].v[j]; }
#pragma unroll
                for (int y = 0; y < NT_YT; ++y) bj[y] = (double)b[y].v[j];
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int y = 0; y < NT_YT; ++y) {
                        accE[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(aj[x], bj[y], accE[x][y], 0, 0, 0);
                        accD[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(dj[x], bj[y], accD[x][y], 0, 0, 0);
                    }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's pieces of the next slab have landed in LDS
        __syncthreads();
    };
    dma(bufA, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int g0 = 0; g0 < G; g0 += 2 * KSL) {
        slab(bufA, bufB, g0);
        if (g0 + KSL < G) slab(bufB, bufA, g0 + KSL);
    }
    nt_epilogue<T>(accE, accD, sums, rm, flags, flag_pitch, C, G, cell0, C_out, ld_rm, accumulate, c0, i0, wi, wj, lrow, lcol);
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The products of perform_PCA that contract over the GENES (velocyto/analysis.py:678-702 through sklearn's PCA: the projection of the centred
// matrix on a thin block, `transform`, and - with fewer cells than genes - the cells' own Gram matrix):
//     out[i][j] = sum_g A[i][g] B[j][g]  -  rc[i]  -  cc[j]  +  c0            A (M, lda) of TA, B (N, ldb) of TB, out (M, ldo) fp64
// Both operands are walked along their contiguous dimension ("NT"), exactly the shape of the linear all-pairs kernel above, and the kernel is
// that kernel's LDS-DMA form with one product instead of two: 128 x 64 tiles, slabs of 128-byte A rows DMA'd from global memory straight into
// LDS (global_load_lds_dwordx4), XOR-swizzled pieces, f64 matrix cores.  Centring is algebra, not a pass over the matrix: with m the gene means,
//     (X - m) Z         = X Z - 1 (m Z)                     cc = m Z                                  (subspace pass, transform)
//     (X - m)(X - m)^T  = X X^T - a 1^T - 1 a^T + m.m       rc = cc = a = X m, c0 = m.m               (fewer cells than genes)
// so the rows are read AS STORED (f32 or f64; no fp64 block copies of X) and once.  B may be fp64 beside an f32 A (the thin block of the
// subspace iteration must not be rounded to f32): its slab rows are then 256 bytes, 16 pieces, swizzled over 16 rows.
// Thin blocks (N <= 64: one column of tiles) make this a stream over A: 12 GB of f64 rows per pass at 50 000 x 30 000.
// TM: rows of A per workgroup (128: square products, every staged f64 of A feeds 4 matrix instructions per wave; 64: thin blocks, twice the
// workgroups).  NPA: 16-byte pieces of an A row per slab (8: 128-byte rows; 16: 256-byte rows - a thin product is a stream over A, and HBM serves
// 50 000 interleaved row streams the better the longer each request run is).  NST LDS buffers, slab s + NST - 1 requested before slab s is
// multiplied: a wave waits only for the slab it needs next (s_waitcnt vmcnt(<its requests still allowed in flight>)), never for the one just issued.
template <typename TA, typename TB, int TM, int NPA, int NST>
__global__ __launch_bounds__(GM_THREADS) void k_gemm_nt_dma(const TA *__restrict__ A, const TB *__restrict__ B, const double *__restrict__ rc,
                                                            const double *__restrict__ cc, double c0, double *__restrict__ out, int M, int N, int K,
                                                            int64_t lda, int64_t ldb, int64_t ldo, int ntn)
{
    static_assert((TM == 128 || TM == 64) && (NPA == 8 || NPA == 16) && (NST == 2 || NST == 3), "shapes");
    constexpr int TNB = 64;
    constexpr int XT = TM / 32;                                       // 16-row MFMA tiles per wave along i (waves 2 x 2: a wave owns TM / 2 x 32)
    constexpr int PGA = 16 / (int)sizeof(TA), PGB = 16 / (int)sizeof(TB);  // genes per 16-byte piece
    constexpr int KSL = NPA * PGA;                                    // genes per slab
    constexpr int NPB = KSL / PGB;                                    // pieces of a B row of a slab: NPA (same type) or 2 NPA (f32 A, f64 B)
    constexpr int NB = PGA / PGB;                                     // B pieces beside one A piece
    constexpr int ROWB_A = NPA * 16, ROWB_B = NPB * 16;
    constexpr int BUF = TM * ROWB_A + TNB * ROWB_B;
    static_assert(NST * BUF <= 160 * 1024, "the slabs in flight must fit the LDS of a CU");
    __shared__ __attribute__((aligned(16))) char buf0[BUF];           // (separate arrays: see k_cdc_full_linear_dma)
    __shared__ __attribute__((aligned(16))) char buf1[BUF];
    __shared__ __attribute__((aligned(16))) char buf2[NST == 3 ? BUF : 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int total = (int)gridDim.x, per = (total + 7) / 8;
    const int q = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    const int mt = q / ntn, nt = q - mt * ntn;
    if (mt * TM >= M) return;
    const int i0 = mt * TM, j0 = nt * TNB;
    // DMA roles.  One instruction = 64 pieces = 64 / NP rows of NP pieces; lane l writes LDS slot l of the group and reads the genes the swizzle
    // puts there (piece p of row r holds the genes of piece p ^ (r & min(NP, 16) - 1): a fragment read - 16 rows, one piece each - then takes
    // different bank groups in every 8-lane phase).  A wave covers its TM / 4 rows of A and its 16 rows of B.
    constexpr int RIA = 64 / NPA, UA = (TM / 4) / RIA;                // rows per A instruction, A instructions per wave
    constexpr int RIB = 64 / NPB > 0 ? 64 / NPB : 1, LPB = NPB > 64 ? 64 : NPB;
    static_assert(NPB <= 32, "B rows of at most 512 bytes");
    constexpr int UB = 16 / RIB;                                      // B instructions per wave
    constexpr int SWA = NPA - 1, SWB = (NPB > 16 ? 16 : NPB) - 1;    // swizzle masks
    const TA *srcA[UA];
#pragma unroll
    for (int u = 0; u < UA; ++u) {
        const int r = wave * (TM / 4) + u * RIA + lane / NPA;
        srcA[u] = A + (int64_t)min(i0 + r, M - 1) * lda + PGA * ((lane % NPA) ^ (r & SWA));
    }
    const TB *srcB[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
        const int r = wave * 16 + u * RIB + lane / LPB;
        srcB[u] = B + (int64_t)min(j0 + r, N - 1) * ldb + PGB * ((lane % LPB) ^ (r & SWB));
    }
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    auto dma = [&](char *base, int g0) {
#pragma unroll
        for (int u = 0; u < UA; ++u)
            __builtin_amdgcn_global_load_lds((glb_void *)(srcA[u] + g0), (lds_void *)(base + (wave * (TM / 4) + u * RIA) * ROWB_A), 16, 0, 0);
#pragma unroll
        for (int u = 0; u < UB; ++u)
            __builtin_amdgcn_global_load_lds((glb_void *)(srcB[u] + g0), (lds_void *)(base + TM * ROWB_A + (wave * 16 + u * RIB) * ROWB_B), 16, 0, 0);
    };
    const int wm = wave >> 1, wn = wave & 1;
    const int wi = wm * (TM / 2), wj = wn * 32;
    v4d_t acc[XT][2];
#pragma unroll
    for (int x = 0; x < XT; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = v4d_t{0.0, 0.0, 0.0, 0.0};
    const int lrow = lane >> 4, lcol = lane & 15;
    // one slab: slab s + NST - 1 streams into `ahead` (free since the barrier that ended slab s - 1) while the matrix instructions read `cur`
    auto slab = [&](const char *cur, char *ahead, int g0) {
        const bool more = g0 + (NST - 1) * KSL < K;
        if (more) dma(ahead, g0 + (NST - 1) * KSL);
        __builtin_amdgcn_sched_barrier(0);
        const char *as = cur, *bs = cur + TM * ROWB_A;
#pragma unroll
        for (int kk2 = 0; kk2 < NPA / 4; ++kk2) {                      // lane group lrow reads A piece 4 kk2 + lrow and the B pieces beside it
            struct alignas(16) PA { TA v[PGA]; };
            struct alignas(16) PB { TB v[PGB]; };
            PA a[XT];
            PB b[2][NB];
            const int pa = kk2 * 4 + lrow;
#pragma unroll
            for (int x = 0; x < XT; ++x) a[x] = *reinterpret_cast<const PA *>(as + (wi + x * 16 + lcol) * ROWB_A + ((pa ^ (lcol & SWA)) * 16));
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int h = 0; h < NB; ++h)
                    b[y][h] = *reinterpret_cast<const PB *>(bs + (wj + y * 16 + lcol) * ROWB_B + (((pa * NB + h) ^ (lcol & SWB)) * 16));
#pragma unroll
            for (int j = 0; j < PGA; ++j) {
                double aj[XT], bj[2];
#pragma unroll
                for (int x = 0; x < XT; ++x) aj[x] = (double)a[x].v[j];
#pragma unroll
                for (int y = 0; y < 2; ++y) bj[y] = (double)b[y][j / PGB].v[j % PGB];
#pragma unroll
                for (int x = 0; x < XT; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(aj[x], bj[y], acc[x][y], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // this wave's pieces of slab s + 1 have landed (requested before those of the later slabs, and loads retire in order)
        if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * (UA + UB)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    dma(buf0, 0);
    if (NST == 3 && KSL < K) dma(buf1, KSL);
    if (NST == 3 && KSL < K) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(UA + UB) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (NST == 3) {
        for (int g0 = 0; g0 < K; g0 += 3 * KSL) {
            slab(buf0, buf2, g0);
            if (g0 + KSL < K) slab(buf1, buf0, g0 + KSL);
            if (g0 + 2 * KSL < K) slab(buf2, buf1, g0 + 2 * KSL);
        }
    } else {
        for (int g0 = 0; g0 < K; g0 += 2 * KSL) {
            slab(buf0, buf1, g0);
            if (g0 + KSL < K) slab(buf1, buf0, g0 + KSL);
        }
    }
#pragma unroll
    for (int y = 0; y < 2; ++y) {
        const int j = j0 + wj + y * 16 + lcol;
        const double ccj = (cc && j < N) ? cc[j] : 0.0;
#pragma unroll
        for (int x = 0; x < XT; ++x)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + wi + x * 16 + lrow + 4 * r;
                if (i < M && j < N) out[(int64_t)i * ldo + j] = acc[x][y][r] - (rc ? rc[i] : 0.0) - ccj + c0;
            }
    }
}

}  // namespace vcy

using namespace vcy;

extern "C" size_t vcy_gram_workspace_bytes(int64_t C, int64_t G, int64_t L, int symmetric)
{
    DevInfo dev;
    if (device_info(&dev)) dev.cus = 256;
    const int tn = gram_tile_cols(L, symmetric != 0);
    const int64_t nta = (G + GM_T - 1) / GM_T, ntb = (L + tn - 1) / tn;
    const int64_t ntile = symmetric ? nta * (nta + 1) / 2 : nta * ntb;
    const int ksplit = gram_ksplit(ntile, C, G, symmetric ? G : L, dev.cus);
    const size_t means = (size_t)GM_MEAN_BLOCKS * (size_t)G * sizeof(double);
    const size_t parts = ksplit > 1 ? (size_t)ksplit * (size_t)G * (size_t)(symmetric ? G : L) * sizeof(double) : 0;
    return parts > means ? parts : means;
}

extern "C" int vcy_col_means(const void *X, double *mean, void *workspace, int64_t C, int64_t G, int64_t ld, int dtype, vcy_stream stream)
{
    VCY_REQUIRE(X && mean && workspace, "col_means: null pointer");
    VCY_REQUIRE(C > 0 && G > 0 && ld >= G, "col_means: bad shape");
    VCY_REQUIRE(dtype == VCY_F32 || dtype == VCY_F64, "col_means: bad dtype");
    hipStream_t st = as_stream(stream);
    const int nb = (int)(C < GM_MEAN_BLOCKS ? C : GM_MEAN_BLOCKS);
    dim3 grid((unsigned)((G + 255) / 256), (unsigned)nb);
    if (dtype == VCY_F32) hipLaunchKernelGGL(k_col_sums_partial<float>, grid, dim3(256), 0, st, (const float *)X, (double *)workspace, (int)C, (int)G, ld, nb);
    else hipLaunchKernelGGL(k_col_sums_partial<double>, grid, dim3(256), 0, st, (const double *)X, (double *)workspace, (int)C, (int)G, ld, nb);
    VCY_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_col_means_fold, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, st, (const double *)workspace, mean, (int)C, (int)G, nb);
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}

static int check_gram_args(const char *who, const void *X, const void *out, int64_t C, int64_t G, int64_t ld, int64_t ldo, int64_t L, int dtype)
{
    if (!(X && out)) return fail(VCY_ERR_INVALID, "%s: null pointer", who);
    if (!(C > 0 && G > 0 && L > 0 && ld >= G && ldo >= L)) return fail(VCY_ERR_INVALID, "%s: bad shape", who);
    if (C >= (1LL << 31) || G >= (1LL << 31) || L >= (1LL << 31)) return fail(VCY_ERR_INVALID, "%s: dimension too large", who);
    if (!(dtype == VCY_F32 || dtype == VCY_F64)) return fail(VCY_ERR_INVALID, "%s: bad dtype", who);
    if (ld % (dtype == VCY_F32 ? 4 : 2) != 0 || ((uintptr_t)X % 16)) return fail(VCY_ERR_INVALID, "%s: rows of X must be 16-byte aligned", who);
    return VCY_OK;
}

extern "C" int vcy_gram(const void *X, const double *mean, double *gram, void *workspace, int64_t C, int64_t G, int64_t ld, int64_t ldg, int dtype,
                        vcy_stream stream)
{
    int rc = check_gram_args("gram", X, gram, C, G, ld, ldg, G, dtype);
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    if (dtype == VCY_F32) return launch_gram<float, float, true>(X, X, mean, mean, gram, workspace, C, G, G, ld, ld, ldg, st);
    return launch_gram<double, double, true>(X, X, mean, mean, gram, workspace, C, G, G, ld, ld, ldg, st);
}

extern "C" int vcy_gram_tn(const void *X, const double *mean, const double *Y, double *out, void *workspace, int64_t C, int64_t G, int64_t L, int64_t ld,
                           int64_t ldy, int64_t ldo, int dtype, vcy_stream stream)
{
    int rc = check_gram_args("gram_tn", X, out, C, G, ld, ldo, L, dtype);
    if (rc) return rc;
    VCY_REQUIRE(Y && ldy >= L && ldy % 2 == 0 && ((uintptr_t)Y % 16) == 0, "gram_tn: Y must be (C, ldy) fp64 with 16-byte aligned rows");
    hipStream_t st = as_stream(stream);
    if (dtype == VCY_F32) return launch_gram<float, double, false>(X, Y, mean, nullptr, out, workspace, C, G, L, ld, ldy, ldo, st);
    return launch_gram<double, double, false>(X, Y, mean, nullptr, out, workspace, C, G, L, ld, ldy, ldo, st);
}

/* out (M, ldo) fp64 = A B^T - rc 1^T - 1 cc^T + c0 over K genes; see k_gemm_nt_dma */
extern "C" int vcy_gemm_nt(const void *A, const void *B, const double *row_corr, const double *col_corr, double c0, double *out, int64_t M, int64_t N,
                           int64_t K, int64_t lda, int64_t ldb, int64_t ldo, int dtype_a, int dtype_b, vcy_stream stream)
{
    VCY_REQUIRE(A && B && out, "gemm_nt: null pointer");
    VCY_REQUIRE(M > 0 && N > 0 && K > 0 && ldo >= N, "gemm_nt: bad shape");
    VCY_REQUIRE(M < (1LL << 31) && N < (1LL << 31) && K < (1LL << 31), "gemm_nt: dimension too large");
    VCY_REQUIRE((dtype_a == VCY_F32 || dtype_a == VCY_F64) && (dtype_b == dtype_a || dtype_b == VCY_F64), "gemm_nt: A is f32 or f64, B of A's type or f64");
    const int64_t ksl = dtype_a == VCY_F32 ? 32 : 16;                 // genes per slab
    const int64_t kpad = (K + ksl - 1) / ksl * ksl;
    VCY_REQUIRE(lda >= kpad && ldb >= kpad, "gemm_nt: both row pitches must hold whole slabs of 16 (f64 A) / 32 (f32 A) genes, zero beyond K");
    VCY_REQUIRE(lda % (dtype_a == VCY_F32 ? 4 : 2) == 0 && ldb % (dtype_b == VCY_F32 ? 4 : 2) == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0,
                "gemm_nt: rows must be 16-byte aligned");
    // the tile shapes of k_gemm_nt_dma: rows of A per workgroup x bytes of a slab row x LDS buffers
    const int64_t ksl2 = 2 * ksl;
    const bool wide_ok = lda >= (K + ksl2 - 1) / ksl2 * ksl2 && ldb >= (K + ksl2 - 1) / ksl2 * ksl2;
    const int shape = env_int("VCY_GEMM_NT_SHAPE", -1);            // (A/B: 0 = 128 rows x 128 B x 2 buffers, 1 = 64 x 128 x 3, 2 = 64 x 256 x 2, 3 = 64 x 256 x 3)
    // measured at 50 000 x 30 000 (profiles/r06_pca_products.txt): a 50-column block 4.6 / 4.7 / 4.8 / 5.0 ms in shapes 0 / 1 / 2 / 3 (f64 rows; f32 rows
    // 4.1 / 4.1 / 4.6 / 4.6) - 2.6 TB/s of A, twice the issue time of its matrix instructions, whatever the tile: shape 0; two columns of tiles (100
    // columns) 8.9 / 7.6 / 7.9 / 7.9: shape 1
    const int pick = shape >= 0 ? shape : ((N > 64 && N <= 128) ? 1 : 0);
    VCY_REQUIRE(pick <= 1 || wide_ok, "gemm_nt: VCY_GEMM_NT_SHAPE asks for 256-byte slab rows the pitches do not hold");
    const int tm = pick == 0 ? 128 : 64;
    const int64_t ntm = (M + tm - 1) / tm, ntn = (N + 63) / 64;
    const int64_t blocks = (ntm * ntn + 7) / 8 * 8;
    VCY_REQUIRE(blocks < (1LL << 31), "gemm_nt: grid too large");
    hipStream_t st = as_stream(stream);
#define VCY_GNT_L(TA, TB, TM_, NPA_, NST_)                                                                                                 \
    hipLaunchKernelGGL((k_gemm_nt_dma<TA, TB, TM_, NPA_, NST_>), dim3((unsigned)blocks), dim3(GM_THREADS), 0, st, (const TA *)A, (const TB *)B, row_corr, \
                       col_corr, c0, out, (int)M, (int)N, (int)K, lda, ldb, ldo, (int)ntn)
#define VCY_GNT(TA, TB)                                                                                                                    \
    do {                                                                                                                                   \
        if (pick == 0) VCY_GNT_L(TA, TB, 128, 8, 2);                                                                                       \
        else if (pick == 1) VCY_GNT_L(TA, TB, 64, 8, 3);                                                                                   \
        else if (pick == 2) VCY_GNT_L(TA, TB, 64, 16, 2);                                                                                  \
        else VCY_GNT_L(TA, TB, 64, 16, 3);                                                                                                 \
    } while (0)
    if (dtype_a == VCY_F64) VCY_GNT(double, double);
    else if (dtype_b == VCY_F64) VCY_GNT(float, double);
    else VCY_GNT(float, float);
#undef VCY_GNT_L
#undef VCY_GNT
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}

// workspace of the linear all-pairs variant: five f64 sums per cell, then the repair flags - one uint16 per output row and 16 columns
static inline int64_t nt_flag_pitch(int64_t C) { return (C + NT_N - 1) / NT_N * (NT_N / 16); }
static inline size_t nt_sums_bytes(int64_t C) { return ((size_t)C * 5 * sizeof(double) + 255) / 256 * 256; }

extern "C" size_t vcy_coldeltacor_full_linear_workspace_bytes(int64_t C, int64_t C_out)
{
    if (C <= 0 || C_out <= 0) return 0;
    return nt_sums_bytes(C) + (size_t)C_out * (size_t)nt_flag_pitch(C) * sizeof(unsigned short);
}

extern "C" int vcy_coldeltacor_full_linear(const void *e, const void *d, void *rm, void *workspace, int64_t C, int64_t G, int64_t ld, int64_t cell0,
                                           int64_t C_out, int64_t ld_rm, int accumulate, int dtype, vcy_stream stream)
{
    VCY_REQUIRE(e && d && rm && workspace, "coldeltacor_full_linear: null pointer");
    VCY_REQUIRE(C > 0 && G > 0 && C_out > 0 && cell0 >= 0 && cell0 + C_out <= C && ld >= G && ld_rm >= C, "coldeltacor_full_linear: bad shape");
    VCY_REQUIRE(C < (1LL << 31) && G < (1LL << 31), "coldeltacor_full_linear: dimension too large");
    VCY_REQUIRE(dtype == VCY_F32 || dtype == VCY_F64, "coldeltacor_full_linear: bad dtype");
    VCY_REQUIRE(ld % NT_KS == 0 && ((uintptr_t)e % 16) == 0 && ((uintptr_t)d % 16) == 0,
                "coldeltacor_full_linear: the row pitch must be a multiple of 16 elements (zero beyond G) and the matrices 16-byte aligned");
    VCY_REQUIRE((uintptr_t)workspace % 8 == 0, "coldeltacor_full_linear: the workspace must be 8-byte aligned");
    hipStream_t st = as_stream(stream);
    double *sums = (double *)workspace;
    unsigned short *flags = (unsigned short *)((char *)workspace + nt_sums_bytes(C));
    const int64_t fp = nt_flag_pitch(C);
    const int64_t ntm = (C_out + NT_M - 1) / NT_M, ntn = (C + NT_N - 1) / NT_N;
    const int64_t blocks = (ntm * ntn + 7) / 8 * 8;
    VCY_REQUIRE(blocks < (1LL << 31), "coldeltacor_full_linear: grid too large");
    const size_t lds = (size_t)2 * (2 * NT_M + NT_N) * NT_LD * sizeof(double);
    int rc;
#define VCY_NT(T)                                                                                                                          \
    do {                                                                                                                                   \
        hipLaunchKernelGGL(k_cell_linear_sums<T>, dim3((unsigned)C), dim3(256), 0, st, (const T *)e, (const T *)d, sums, (int)G, ld, cell0, (int)C_out); \
        VCY_LAUNCH_CHECK();                                                                                                                \
        auto kern = k_cdc_full_linear<T, T>;                                                                                               \
        rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds);                                                                \
        if (rc) return rc;                                                                                                                 \
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(GM_THREADS), lds, st, (const T *)e, (const T *)d, (const double *)sums, (T *)rm, flags, fp, \
                           (int)C, (int)G, ld, cell0, (int)C_out, ld_rm, accumulate, (int)ntn);                                            \
    } while (0)
#define VCY_NT_DMA_LAUNCH(T)                                                                                                               \
    do {                                                                                                                                   \
        hipLaunchKernelGGL(k_cell_linear_sums<T>, dim3((unsigned)C), dim3(256), 0, st, (const T *)e, (const T *)d, sums, (int)G, ld, cell0, (int)C_out); \
        VCY_LAUNCH_CHECK();                                                                                                                \
        hipLaunchKernelGGL(k_cdc_full_linear_dma<T>, dim3((unsigned)blocks), dim3(GM_THREADS), 0, st, (const T *)e, (const T *)d, (const double *)sums, \
                           (T *)rm, flags, fp, (int)C, (int)G, ld, cell0, (int)C_out, ld_rm, accumulate, (int)ntn);                        \
    } while (0)
#define VCY_NT_REPAIR(T)                                                                                                                   \
    hipLaunchKernelGGL(k_cdc_linear_repair<T>, dim3((unsigned)C_out), dim3(256), 0, st, (const T *)e, (const T *)d, (const unsigned short *)flags, fp, \
                       (T *)rm, (int)C, (int)G, ld, cell0, ld_rm, accumulate)
    // the DMA form needs whole 128-byte slab rows inside the pitch (ld a multiple of 16 f64 / 32 f32 elements: the layout's 64-element padding
    // gives both); VCY_NT_DMA=0 runs the register-staged form (A/B)
    const bool dma_ok = env_int("VCY_NT_DMA", 1) != 0 && ld % (dtype == VCY_F32 ? 32 : 16) == 0;
    if (dtype == VCY_F32) { if (dma_ok) VCY_NT_DMA_LAUNCH(float); else VCY_NT(float); }
    else { if (dma_ok) VCY_NT_DMA_LAUNCH(double); else VCY_NT(double); }
    VCY_LAUNCH_CHECK();
    // the pairs the expansion cannot carry (nearly identical cells, duplicates, a nearly constant d_c), in the reference's centred form
    if (dtype == VCY_F32) VCY_NT_REPAIR(float); else VCY_NT_REPAIR(double);
#undef VCY_NT_REPAIR
#undef VCY_NT_DMA_LAUNCH
#undef VCY_NT
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}
