// knn.hip -- stage A: exact Euclidean kNN search + the greedy balanced-kNN selection.
//
// Reference: sklearn.neighbors.NearestNeighbors as called by neighbors.knn_distance_matrix
// (neighbors.py:363-376), BalancedKNN.fit/kneighbors (:239-243, 282) and
// estimate_transition_prob (analysis.py:1547-1549); numba loops balance_knn_loop[_constrained]
// (neighbors.py:11-140).
//
// Search: a workgroup owns QB = 8 query cells.
//   phase 1  fp32 squared distances to all C candidates, lanes over candidates reading the
//            feature-major (P, C) copy of the space (coalesced), 8 accumulators per lane;
//            the (8, C) distance rows go to a workspace that stays L2-resident for phase 2.
//   phase 2  per query: find a threshold T with count(row <= T) >= Ksel = k + margin:
//            fast path - every thread kept the two smallest distances of its strided slice in
//            phase 1; the Ksel-th smallest of those 512 local minima bounds the true Ksel-th
//            from above and is tight (one 512-element bitonic sort in LDS, no atomics on the
//            row); fallback - MSB-first radix select on the 64-bit key (sortable distance bits
//            << 32 | index) giving exactly Ksel keys.  Then gather the candidates <= T,
//            recompute their distances exactly in fp64, bitonic-sort (fp64 distance, index) in
//            LDS and emit the first k.  Both paths are deterministic, ties by index.
// ROWS = false (Ksel <= 512, C <= 2^18; the common case) never materialises the distance rows: in phase 1 every thread
// keeps the FOUR smallest distances of its strided slice per query as sortable words (distance bits with the low `nb`
// mantissa bits replaced by the slice position), phase 2 takes the threshold from the two smallest of every thread as
// before and then collects the tracked words within the threshold (compared on the high bits, one step of slack, so
// everything the exact comparison would admit is in) - no row write, no row scan.  A thread whose fourth word is within
// the threshold may hold more: that query alone recomputes its row and takes the radix-select path.
// The exact distances are sums of ROUNDED squares in feature order (no fused multiply-add): two candidates whose
// displacements are permutations / sign flips of each other then tie exactly, as they do in a plain C or numpy loop.
// The fp32 pass only has to get the candidate SET right (margin of 8 near-ties); order and
// returned distances are fp64, matching the reference's fp64 search up to exact ties (which
// sklearn orders arbitrarily and this kernel orders by index; with include_self the query is an
// ordinary candidate at distance 0).
#include <math.h>
#include "common.h"

// No implicit fused multiply-adds in this file (hipcc contracts a*b+c by default, even through __dmul_rn / __dadd_rn):
// the exact fp64 distances must be sums of ROUNDED squares so that mathematically tied candidates tie bit for bit, as
// they do in a plain C / numpy loop.  The fp32 distance pass asks for its FMAs explicitly (fmaf).
#pragma clang fp contract(off)

namespace vcy {

constexpr int KNN_QB = 8;
constexpr int KNN_MAXSEL = 4096;
constexpr int KNN_MARGIN = 8;
constexpr int KNN_NC = 4;          // candidates per thread per pass of the distance loop

__device__ __forceinline__ uint32_t f32_key(float f)
{
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// LARGE = false: candidate arrays (fp64 distance, index)[nsort] live in LDS (k <= ~4k);
// LARGE = true : they live in a per-workgroup slice of the global workspace (any k < C), same code.
template <bool LARGE, bool ROWS>
__global__ __launch_bounds__(256) void k_knn_search(const float *__restrict__ xt, const double *__restrict__ x64, const float *__restrict__ qt,
                                                     const double *__restrict__ q64, int64_t ldq, int32_t *__restrict__ idx_out,
                                                     double *__restrict__ dist_out, float *__restrict__ ws, char *__restrict__ ws_sort, int C, int P,
                                                     int64_t ldx, int64_t q0, int Q, int k, int ksel, int nsort, int include_self, int nb)
{
    // qt/q64 == NULL: the queries are rows q0.. of the point set itself (kneighbors / kneighbors_graph of the fitted
    // data); otherwise qt (P, ldq) / q64 (Q_total, P) hold EXTERNAL query points (NearestNeighbors.kneighbors(X)) and
    // no candidate is excluded.
    const bool external = qt != nullptr;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *sd;
    int *si;
    float *xq;
    if (LARGE) {
        char *mine = ws_sort + (size_t)blockIdx.x * (size_t)nsort * (sizeof(double) + sizeof(int));
        sd = reinterpret_cast<double *>(mine);
        si = reinterpret_cast<int *>(sd + nsort);
        xq = reinterpret_cast<float *>(smem);
    } else {
        sd = reinterpret_cast<double *>(smem);                        // [nsort] fp64 distances
        si = reinterpret_cast<int *>(sd + nsort);                     // [nsort] indices
        xq = reinterpret_cast<float *>(si + nsort);                   // [P][QB] (16-byte aligned: nsort is a multiple of 128)
    }
    __shared__ unsigned hist[256];
    __shared__ float cand[512];
    __shared__ unsigned long long s_prefix;
    __shared__ unsigned s_rank, s_count, s_over;
    __shared__ int s_overtid[8];
    const int tid = threadIdx.x;
    const int qb0 = blockIdx.x * KNN_QB;
    const int nq = min(KNN_QB, Q - qb0);

    for (int t = tid; t < KNN_QB * P; t += 256) {
        const int p = t / KNN_QB, qq = t - p * KNN_QB;           // [P][QB]: the 8 query coordinates of one feature are 32 contiguous bytes
        xq[t] = qq < nq ? (external ? qt[(int64_t)p * ldq + q0 + qb0 + qq] : xt[(int64_t)p * ldx + q0 + qb0 + qq]) : 0.f;
    }
    __syncthreads();
    // ---- phase 1: distances
    // ROWS: the (QB, C) distance rows of this workgroup; !ROWS: one scratch row per workgroup (fallback of a single query)
    float *wrow = ROWS ? ws + (int64_t)qb0 * C : ws + (int64_t)blockIdx.x * C;
    float m0[KNN_QB], m1[KNN_QB];       // ROWS: two smallest distances this thread has seen, per query
    unsigned tk[KNN_QB][4];             // !ROWS: four smallest (distance bits | slice position) words, ascending
    const unsigned lowmask = (1u << nb) - 1u;
#pragma unroll
    for (int qq = 0; qq < KNN_QB; ++qq) {
        m0[qq] = INFINITY; m1[qq] = INFINITY;
#pragma unroll
        for (int l = 0; l < 4; ++l) tk[qq][l] = 0xffffffffu;
    }
    // KNN_NC candidates (j, j + 256, ...) per thread and iteration: the LDS reads of the 8 query coordinates are
    // shared by all of them, and KNN_NC independent global loads are in flight per feature
    for (int j = tid; j < C; j += 256 * KNN_NC) {
        float acc[KNN_NC][KNN_QB];
#pragma unroll
        for (int c = 0; c < KNN_NC; ++c)
#pragma unroll
            for (int qq = 0; qq < KNN_QB; ++qq) acc[c][qq] = 0.f;
#pragma unroll 4
        for (int p = 0; p < P; ++p) {
            float xv[KNN_NC];
#pragma unroll
            for (int c = 0; c < KNN_NC; ++c) xv[c] = (j + 256 * c < C) ? xt[(int64_t)p * ldx + j + 256 * c] : 0.f;
            float qv8[KNN_QB];
            *reinterpret_cast<float4 *>(&qv8[0]) = *reinterpret_cast<const float4 *>(&xq[p * KNN_QB]);       // two broadcast ds_read_b128
            *reinterpret_cast<float4 *>(&qv8[4]) = *reinterpret_cast<const float4 *>(&xq[p * KNN_QB + 4]);
#pragma unroll
            for (int qq = 0; qq < KNN_QB; ++qq) {
#pragma unroll
                for (int c = 0; c < KNN_NC; ++c) { const float df = qv8[qq] - xv[c]; acc[c][qq] = fmaf(df, df, acc[c][qq]); }
            }
        }
        if (!ROWS) {
#pragma unroll
            for (int c = 0; c < KNN_NC; ++c) {
                const int jc = j + 256 * c;
                const unsigned pos = (unsigned)(jc - tid) >> 8;               // slice position: jc = tid + 256 * pos
#pragma unroll
                for (int qq = 0; qq < KNN_QB; ++qq) {
                    const float v = acc[c][qq];
                    const bool self = !include_self && !external && (int64_t)jc == q0 + qb0 + qq;
                    // non-negative floats order like their bit patterns; out of range / excluded / non-finite -> never tracked
                    unsigned u = (jc < C && qq < nq && !self && v < INFINITY) ? ((__float_as_uint(v) & ~lowmask) | pos) : 0xffffffffu;
                    unsigned a = min(tk[qq][0], u); u = max(tk[qq][0], u); tk[qq][0] = a;
                    a = min(tk[qq][1], u); u = max(tk[qq][1], u); tk[qq][1] = a;
                    a = min(tk[qq][2], u); u = max(tk[qq][2], u); tk[qq][2] = a;
                    tk[qq][3] = min(tk[qq][3], u);
                }
            }
            continue;
        }
#pragma unroll
        for (int c = 0; c < KNN_NC; ++c) {
            const int jc = j + 256 * c;
            if (jc < C) {
#pragma unroll
                for (int qq = 0; qq < KNN_QB; ++qq) {
                    if (qq < nq) {
                        float v = acc[c][qq];
                        if (!include_self && !external && (int64_t)jc == q0 + qb0 + qq) v = INFINITY;   // query excluded (kneighbors_graph(X=None))
                        wrow[(int64_t)qq * C + jc] = v;
                        if (v < m1[qq]) {
                            if (v < m0[qq]) { m1[qq] = m0[qq]; m0[qq] = v; }
                            else m1[qq] = v;
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    // ---- phase 2: select + exact re-rank, one query at a time
    for (int qq = 0; qq < nq; ++qq) {
        const float *row = ROWS ? wrow + (int64_t)qq * C : wrow;
        const int64_t qcell = q0 + qb0 + qq;
        unsigned long long prefix = 0;
        bool have_threshold = false;
        if (!ROWS) {
            unsigned a0 = tk[0][0], a1 = tk[0][1], a2 = tk[0][2], a3 = tk[0][3];
#pragma unroll
            for (int t = 1; t < KNN_QB; ++t) { if (qq == t) { a0 = tk[t][0]; a1 = tk[t][1]; a2 = tk[t][2]; a3 = tk[t][3]; } }
            unsigned *candu = reinterpret_cast<unsigned *>(cand);
            candu[2 * tid] = a0;
            candu[2 * tid + 1] = a1;
            __syncthreads();
            for (int size = 2; size <= 512; size <<= 1) {
                for (int stride = size >> 1; stride > 0; stride >>= 1) {
                    const int lo = 2 * tid - (tid & (stride - 1)), hi = lo + stride;
                    const bool up = ((lo & size) == 0);
                    const unsigned a = candu[lo], b = candu[hi];
                    if ((a > b) == up) { candu[lo] = b; candu[hi] = a; }
                    __syncthreads();
                }
            }
            const unsigned T = candu[ksel - 1];                 // >= the Ksel-th smallest word overall
            __syncthreads();
            if (T < 0x7f800000u) {
                if (tid == 0) { s_count = 0; s_over = 0; }
                for (int t = tid; t < nsort; t += 256) { sd[t] = INFINITY; si[t] = 0x7fffffff; }
                __syncthreads();
                const unsigned Tm = (T >> nb) + 1u;             // high bits only, one step of slack: a superset of {distance <= T}
                auto take = [&](unsigned a) {
                    if ((a >> nb) <= Tm) {
                        const unsigned pos = atomicAdd(&s_count, 1u);
                        if (pos < (unsigned)nsort) si[pos] = tid + 256 * (int)(a & lowmask);
                    }
                };
                // a thread whose fourth word is within the threshold may hold more (about 0.5 % of the queries have one such
                // thread): it hands its whole slice to the workgroup instead of contributing its four words
                const bool mine_over = (a3 >> nb) <= Tm;
                if (mine_over) {
                    const unsigned o = atomicAdd(&s_over, 1u);
                    if (o < 8u) s_overtid[o] = tid;
                } else {
                    take(a0); take(a1); take(a2); take(a3);
                }
                __syncthreads();
                const unsigned nover = s_over;
                if (nover > 0u && nover <= 8u) {
                    const int nslice = (C + 255) / 256;
                    for (unsigned o = 0; o < nover; ++o) {
                        const int ot = s_overtid[o];
                        for (int i = tid; i < nslice; i += 256) {       // one candidate of the slice per thread
                            const int j = ot + 256 * i;
                            if (j < C && (include_self || external || (int64_t)j != qcell)) {
                                float d = 0.f;
                                for (int p = 0; p < P; ++p) { const float df = xq[p * KNN_QB + qq] - xt[(int64_t)p * ldx + j]; d = fmaf(df, df, d); }
                                if (d < INFINITY && (__float_as_uint(d) >> nb) <= Tm) {
                                    const unsigned pos = atomicAdd(&s_count, 1u);
                                    if (pos < (unsigned)nsort) si[pos] = j;
                                }
                            }
                        }
                    }
                    __syncthreads();
                }
                have_threshold = nover <= 8u && s_count <= (unsigned)nsort;     // >= ksel by construction
                __syncthreads();
            }
            if (!have_threshold) {
                // rare: materialise this query's fp32 distance row (same arithmetic as phase 1) for the radix-select path
                for (int j = tid; j < C; j += 1024) {               // four independent load streams per thread
                    float d[4] = {0.f, 0.f, 0.f, 0.f};
                    for (int p = 0; p < P; ++p) {
                        const float qv = xq[p * KNN_QB + qq];
                        float xv[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) xv[c] = (j + 256 * c < C) ? xt[(int64_t)p * ldx + j + 256 * c] : 0.f;
#pragma unroll
                        for (int c = 0; c < 4; ++c) { const float df = qv - xv[c]; d[c] = fmaf(df, df, d[c]); }
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int jc = j + 256 * c;
                        if (jc < C) wrow[jc] = (!include_self && !external && (int64_t)jc == qcell) ? INFINITY : d[c];
                    }
                }
                __syncthreads();
            }
        }
        if (ROWS && ksel <= 512) {
            // ---- fast path: threshold from the 512 per-thread local minima
            float mm0 = m0[0], mm1 = m1[0];
#pragma unroll
            for (int t = 1; t < KNN_QB; ++t) { if (qq == t) { mm0 = m0[t]; mm1 = m1[t]; } }
            cand[2 * tid] = mm0;
            cand[2 * tid + 1] = mm1;
            __syncthreads();
            for (int size = 2; size <= 512; size <<= 1) {
                for (int stride = size >> 1; stride > 0; stride >>= 1) {
                    const int lo = 2 * tid - (tid & (stride - 1)), hi = lo + stride;
                    const bool up = ((lo & size) == 0);
                    const float a = cand[lo], b = cand[hi];
                    if ((a > b) == up) { cand[lo] = b; cand[hi] = a; }
                    __syncthreads();
                }
            }
            const float T = cand[ksel - 1];
            __syncthreads();
            if (T < INFINITY) {
                if (tid == 0) s_count = 0;
                for (int t = tid; t < nsort; t += 256) { sd[t] = INFINITY; si[t] = 0x7fffffff; }
                __syncthreads();
                for (int j = tid; j < C; j += 256) {
                    if (row[j] <= T) {
                        const unsigned pos = atomicAdd(&s_count, 1u);
                        if (pos < (unsigned)nsort) si[pos] = j;
                    }
                }
                __syncthreads();
                have_threshold = s_count <= (unsigned)nsort;   // >= ksel by construction
                __syncthreads();
            }
        }
        if (!have_threshold) {
        unsigned rank = (unsigned)(ksel - 1);
        for (int pass = 0; pass < 8; ++pass) {
            const int shift = 8 * (7 - pass);
            hist[tid] = 0;
            __syncthreads();
            for (int j = tid; j < C; j += 256) {
                const unsigned long long key = ((unsigned long long)f32_key(row[j]) << 32) | (unsigned)j;
                const bool match = (pass == 0) || ((key >> (shift + 8)) == (prefix >> (shift + 8)));
                if (match) atomicAdd(&hist[(unsigned)((key >> shift) & 0xff)], 1u);
            }
            __syncthreads();
            if (tid < 64) {
                const unsigned h0 = hist[tid * 4], h1 = hist[tid * 4 + 1], h2 = hist[tid * 4 + 2], h3 = hist[tid * 4 + 3];
                const unsigned tot = h0 + h1 + h2 + h3;
                unsigned incl = tot;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const unsigned o = __shfl_up(incl, off, 64);
                    if (tid >= off) incl += o;
                }
                const unsigned excl = incl - tot;
                if (rank >= excl && rank < incl) {
                    unsigned r = rank - excl;
                    int d;
                    if (r < h0) d = 0;
                    else if ((r -= h0) < h1) d = 1;
                    else if ((r -= h1) < h2) d = 2;
                    else { r -= h2; d = 3; }
                    s_prefix = prefix | ((unsigned long long)(tid * 4 + d) << shift);
                    s_rank = r;
                }
            }
            __syncthreads();
            prefix = s_prefix;
            rank = s_rank;
        }
        // gather the ksel keys <= prefix (unordered), exact fp64 distances
        if (tid == 0) s_count = 0;
        for (int t = tid; t < nsort; t += 256) { sd[t] = INFINITY; si[t] = 0x7fffffff; }
        __syncthreads();
        for (int j = tid; j < C; j += 256) {
            const unsigned long long key = ((unsigned long long)f32_key(row[j]) << 32) | (unsigned)j;
            if (key <= prefix) {
                const unsigned pos = atomicAdd(&s_count, 1u);
                if (pos < (unsigned)nsort) si[pos] = j;
            }
        }
        __syncthreads();
        }   // fallback
        const int ncand = have_threshold ? (int)s_count : ksel;
        for (int t = tid; t < ncand; t += 256) {
            const int j = si[t];
            double d2 = 0.0;
            const double *a = (external ? q64 : x64) + qcell * P, *b = x64 + (int64_t)j * P;
            for (int p = 0; p < P; ++p) { const double df = a[p] - b[p]; d2 += df * df; }
            sd[t] = d2;
        }
        __syncthreads();
        // bitonic sort of (sd, si) ascending, ties by index
        for (int size = 2; size <= nsort; size <<= 1) {
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int t = tid; t < nsort / 2; t += 256) {
                    const int lo = 2 * t - (t & (stride - 1));
                    const int hi = lo + stride;
                    const bool up = ((lo & size) == 0);
                    const double a = sd[lo], b = sd[hi];
                    const int ia = si[lo], ib = si[hi];
                    const bool gt = (a > b) || (a == b && ia > ib);
                    if (gt == up) { sd[lo] = b; sd[hi] = a; si[lo] = ib; si[hi] = ia; }
                }
                __syncthreads();
            }
        }
        for (int t = tid; t < k; t += 256) {
            idx_out[(int64_t)(qb0 + qq) * k + t] = si[t];
            const double d2 = sd[t];
            dist_out[(int64_t)(qb0 + qq) * k + t] = sqrt(d2);
        }
        __syncthreads();
    }
}
}  // namespace vcy

using namespace vcy;

static int knn_plan(int64_t C, int64_t k, int include_self, int64_t *ksel_out, int *nsort_out, bool *large_out)
{
    const int64_t avail = include_self ? C : C - 1;
    int64_t ksel = k + KNN_MARGIN;
    if (ksel > avail) ksel = avail;
    int nsort = 128;                        // room for the candidates <= threshold (>= ksel of them)
    while (nsort < 4 * ksel && nsort < KNN_MAXSEL) nsort <<= 1;
    while (nsort < ksel) nsort <<= 1;
    *ksel_out = ksel; *nsort_out = nsort; *large_out = nsort > KNN_MAXSEL;
    return 0;
}

// does this search take the kernel that never materialises the distance rows?
static bool knn_row_free(int64_t C, int64_t ksel, bool large, int *nb_out)
{
    int nb = 1;                                                   // bits for a thread's slice position: jc = tid + 256 * pos
    while (((int64_t)1 << nb) < (C + 255) / 256 + KNN_NC) ++nb;
    if (nb_out) *nb_out = nb;
    // the threshold comes from 2 words per thread: with Ksel near 512 it is loose and most threads overflow
    return !(large || ksel > 128 || nb > 10 || env_int("VCY_KNN_ROWS", 0) == 1);   // VCY_KNN_ROWS=1 forces the row kernel (A/B testing)
}

extern "C" size_t vcy_knn_workspace_bytes(int64_t C, int64_t Q, int64_t k)
{
    const int64_t qpad = (Q + KNN_QB - 1) / KNN_QB * KNN_QB;
    int64_t ksel; int nsort; bool large;
    knn_plan(C, k, 1, &ksel, &nsort, &large);
    // row-free kernel: one scratch row per workgroup (the rare fallback of a single query); otherwise the (Q, C) distance rows
    size_t bytes = (size_t)(knn_row_free(C, ksel, large, nullptr) ? qpad / KNN_QB : qpad) * (size_t)C * sizeof(float);
    if (large) bytes += (size_t)(qpad / KNN_QB) * (size_t)nsort * (sizeof(double) + sizeof(int)) + 256;
    return bytes;
}

extern "C" int vcy_knn_row_free(int64_t C, int64_t k)
{
    int64_t ksel; int nsort; bool large;
    knn_plan(C, k, 1, &ksel, &nsort, &large);
    return knn_row_free(C, ksel, large, nullptr) ? 1 : 0;
}

static int knn_search_impl(const float *xt, const double *x64, const float *qt, const double *q64, int64_t ldq, int32_t *idx, double *dist,
                           void *workspace, int64_t C, int64_t P, int64_t ldx, int64_t q0, int64_t Q, int64_t k, int include_self,
                           vcy_stream stream)
{
    VCY_REQUIRE(xt && x64 && idx && dist && workspace, "knn_search: null pointer");
    VCY_REQUIRE(C > 1 && P > 0 && ldx >= C && Q > 0 && q0 >= 0 && (qt != nullptr || q0 + Q <= C), "knn_search: bad shape");
    const int64_t avail = include_self ? C : C - 1;
    VCY_REQUIRE(k > 0 && k <= avail, "knn_search: k exceeds the number of candidates");
    int64_t ksel; int nsort; bool large;
    knn_plan(C, k, include_self, &ksel, &nsort, &large);
    const int64_t qpad = (Q + KNN_QB - 1) / KNN_QB * KNN_QB;
    char *ws_sort = (char *)workspace + (((size_t)qpad * (size_t)C * sizeof(float) + 255) & ~(size_t)255);      // (large searches always hold the rows)
    const unsigned blocks = (unsigned)((Q + KNN_QB - 1) / KNN_QB);
    const size_t lds = (large ? 0 : (size_t)nsort * (sizeof(double) + sizeof(int))) + (size_t)KNN_QB * P * sizeof(float);
    VCY_REQUIRE(lds <= 150 * 1024, "knn_search: feature dimension too large for LDS");
    int nb = 1;
    const bool rows = !knn_row_free(C, ksel, large, &nb);
#define VCY_KNN_LAUNCH(L, R)                                                                                                       \
    do {                                                                                                                           \
        { const int rc_ = ensure_dynamic_lds(reinterpret_cast<const void *>(k_knn_search<L, R>), lds); if (rc_) return rc_; }    \
        hipLaunchKernelGGL((k_knn_search<L, R>), dim3(blocks), dim3(256), lds, as_stream(stream), xt, x64, qt, q64, ldq, idx, dist, (float *)workspace, \
                           ws_sort, (int)C, (int)P, ldx, q0, (int)Q, (int)k, (int)ksel, nsort, include_self, nb);                \
    } while (0)
    if (large) VCY_KNN_LAUNCH(true, true);
    else if (rows) VCY_KNN_LAUNCH(false, true);
    else VCY_KNN_LAUNCH(false, false);
#undef VCY_KNN_LAUNCH
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}

extern "C" int vcy_knn_search(const float *xt, const double *x64, int32_t *idx, double *dist, void *workspace, int64_t C, int64_t P,
                              int64_t ldx, int64_t q0, int64_t Q, int64_t k, int include_self, vcy_stream stream)
{
    return knn_search_impl(xt, x64, nullptr, nullptr, 0, idx, dist, workspace, C, P, ldx, q0, Q, k, include_self, stream);
}

extern "C" int vcy_knn_query(const float *xt, const double *x64, const float *qt, const double *q64, int64_t ldq, int32_t *idx, double *dist,
                             void *workspace, int64_t C, int64_t P, int64_t ldx, int64_t q0, int64_t Q, int64_t k, vcy_stream stream)
{
    VCY_REQUIRE(qt && q64 && ldq >= q0 + Q, "knn_query: bad query arguments");
    return knn_search_impl(xt, x64, qt, q64, ldq, idx, dist, workspace, C, P, ldx, q0, Q, k, /*include_self*/ 1, stream);
}

// Sequential greedy balancing (neighbors.py:47-69 / 113-137): for each cell `el` in the given
// order take, from its distance-sorted sight list, the first k cells m != el whose in-degree
// l[m] is still below maxl (and, if constrained, share el's group); own index -> column 0;
// if the sight list runs out, pad with el itself at distance dist[el,0] (:65-69).
extern "C" int vcy_balance_knn_host(const int64_t *dsi, const double *dist, const int64_t *lsi, const int64_t *groups, int64_t n,
                                    int64_t K, int64_t maxl, int64_t k, int return_distance, double *dist_new, int64_t *dsi_new,
                                    int64_t *l)
{
    VCY_REQUIRE(dsi && dist && lsi && dist_new && dsi_new && l, "balance_knn: null pointer");
    VCY_REQUIRE(n > 0 && K >= k && k > 0, "balance_knn: sight needs to be bigger than k");
    const int64_t w = k + 1;
    for (int64_t t = 0; t < n * w; ++t) { dsi_new[t] = -1; dist_new[t] = 0.0; }
    for (int64_t t = 0; t < n; ++t) l[t] = 0;
    for (int64_t it = 0; it < n; ++it) {
        const int64_t el = lsi[it];
        if (el < 0 || el >= n) return fail(VCY_ERR_INVALID, "%s: lsi entry out of range", "balance_knn");
        const int64_t *sight = dsi + el * K;
        int64_t taken = 0, j = 0;
        for (; j < K && taken < k; ++j) {
            const int64_t m = sight[j];
            if (m == el) { dsi_new[el * w] = el; continue; }
            if (m < 0 || m >= n) return fail(VCY_ERR_INVALID, "%s: dsi entry out of range", "balance_knn");
            if (groups && groups[m] != groups[el]) continue;
            if (l[m] >= maxl) continue;
            ++taken;
            dsi_new[el * w + taken] = m;
            if (return_distance) dist_new[el * w + taken] = dist[el * K + j];
            ++l[m];
        }
        for (; taken < k;) {   // sight exhausted
            ++taken;
            dsi_new[el * w + taken] = el;
            dist_new[el * w + taken] = dist[el * K];
        }
    }
    if (!return_distance)
        for (int64_t t = 0; t < n * w; ++t) dist_new[t] = 1.0;
    return VCY_OK;
}

// The same loop on the sight lists AS THE DEVICE SEARCH RETURNS THEM: int32 neighbour numbers (a quarter of the host memory of
// the int64 + fp64 pair the reference holds - its default sight is the whole dataset, analysis.py:985-988: (C, C) lists, 40 GB of
// host arrays at 50 000 cells against 10 GB here), no distances on the host at all.  Instead of distances it returns the POSITION
// j of every selected neighbour in its cell's sight list (pos_new, -1 for column 0 and for padded slots), from which the caller
// gathers dist[el, j] where the distances live (the device).  Selection identical to vcy_balance_knn_host.
extern "C" int vcy_balance_knn_host32(const int32_t *dsi, const int64_t *lsi, const int64_t *groups, int64_t n, int64_t K, int64_t maxl,
                                      int64_t k, int32_t *pos_new, int64_t *dsi_new, int64_t *l)
{
    VCY_REQUIRE(dsi && lsi && pos_new && dsi_new && l, "balance_knn32: null pointer");
    VCY_REQUIRE(n > 0 && K >= k && k > 0 && K < (1LL << 31), "balance_knn32: sight needs to be bigger than k");
    const int64_t w = k + 1;
    for (int64_t t = 0; t < n * w; ++t) { dsi_new[t] = -1; pos_new[t] = -1; }
    for (int64_t t = 0; t < n; ++t) l[t] = 0;
    for (int64_t it = 0; it < n; ++it) {
        const int64_t el = lsi[it];
        if (el < 0 || el >= n) return fail(VCY_ERR_INVALID, "%s: lsi entry out of range", "balance_knn32");
        const int32_t *sight = dsi + el * K;
        int64_t taken = 0, j = 0;
        for (; j < K && taken < k; ++j) {
            const int64_t m = sight[j];
            if (m == el) { dsi_new[el * w] = el; continue; }
            if (m < 0 || m >= n) return fail(VCY_ERR_INVALID, "%s: dsi entry out of range", "balance_knn32");
            if (groups && groups[m] != groups[el]) continue;
            if (l[m] >= maxl) continue;
            ++taken;
            dsi_new[el * w + taken] = m;
            pos_new[el * w + taken] = (int32_t)j;
            ++l[m];
        }
        for (; taken < k;) {   // sight exhausted: padded with el itself (distance dist[el, 0], neighbors.py:65-69) - pos stays -1
            ++taken;
            dsi_new[el * w + taken] = el;
        }
    }
    return VCY_OK;
}
