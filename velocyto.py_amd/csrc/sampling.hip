// Host side of estimate_transition_prob's neighbour sampling (analysis.py:1561-1564):
//     sampling_ixs = np.stack(np.random.choice(n, size=size, replace=False, p=p) for every cell)
// with numpy's legacy global RNG.  The reference's results depend on that stream draw by draw, so the facade keeps it; what
// costs 3 s at 50 000 cells is not the random numbers (numpy produces 25 M doubles in 0.15 s) but 50 000 trips through
// RandomState.choice.  This file restates choice(replace=False, p) (numpy/random/mtrand.pyx, unchanged since 1.7) over a
// pool of uniforms the caller drew from the same RandomState in one call - RandomState.random_sample(N) is the same
// sequence as any split into rand(k) calls:
//     n_uniq = 0; p = p.copy()
//     while n_uniq < size:
//         x = rand(size - n_uniq);  p[found[:n_uniq]] = 0
//         cdf = cumsum(p); cdf /= cdf[-1];  new = cdf.searchsorted(x, side="right")
//         new = first occurrences of new, in draw order;  found[n_uniq : n_uniq + len(new)] = new;  n_uniq += len(new)
// cumsum is a sequential fp64 accumulation and the division is elementwise, so the cdf is reproduced bit for bit and every
// searchsorted lands where numpy's does.  Pure host code (no device work): it lives in the library so that the binding of
// INTEGRATION.md covers it.
#include "common.h"
#include <vector>

namespace {
#pragma clang fp contract(off)
}

// pool: `pool_len` uniforms in [0, 1) in stream order.  p: n probabilities (already normalised as the caller passes them to
// numpy).  out: (cells, size) int64.  Returns VCY_OK and sets *cells_done (cells whose draws fitted in the pool) and *consumed
// (uniforms those cells took: the caller advances its RandomState by exactly that many).
extern "C" int vcy_choice_stream_host(const double *pool, int64_t pool_len, const double *p, int64_t n, int64_t size, int64_t cells,
                                      int64_t *out, int64_t *cells_done, int64_t *consumed)
{
#pragma clang fp contract(off)
    VCY_REQUIRE(pool && p && out && cells_done && consumed, "choice_stream: null pointer");
    VCY_REQUIRE(n > 0 && size >= 0 && size <= n && cells >= 0 && pool_len >= 0, "choice_stream: need 0 <= size <= n, n > 0");
    int64_t positive = 0;
    for (int64_t i = 0; i < n; ++i) {
        VCY_REQUIRE(p[i] >= 0.0, "choice_stream: probabilities are not non-negative");          // numpy's messages
        positive += p[i] > 0.0;
    }
    VCY_REQUIRE(positive >= size, "choice_stream: Fewer non-zero entries in p than size");
    int64_t LUT = 128;                                           // power of two: v * LUT and b / LUT are exact; about two buckets
    while (LUT < 2 * n && LUT < 4096) LUT *= 2;                  // per entry, so that the scan below is a few entries long
    constexpr int64_t WMAX = 32;                                 // widest bucket the fixed-length scan of round 1 is used for
    std::vector<double> pw((size_t)n), cs((size_t)n), cdf((size_t)n), cdf0((size_t)(n + WMAX), 2.0);     // cdf0 padded with values no draw reaches
    std::vector<int64_t> stamp((size_t)n, -1);
    std::vector<int32_t> lut0((size_t)LUT + 1);
    // cdf = cumsum(w) / cumsum(w)[-1] exactly as numpy forms it: a sequential fp64 accumulation, then an elementwise division.
    auto cumsum = [&](const double *w, double *c) {
        double acc = 0.0;
        for (int64_t i = 0; i < n; ++i) { acc = acc + w[i]; c[i] = acc; }
    };
    // The first round of every cell sees the untouched p and takes most of the draws: its cdf gets a bucket table over [0, 1),
    // table[b] = first index with cdf > b / LUT, a lower bound of searchsorted(cdf, v, "right") for every v in bucket
    // b = floor(v * LUT); the answer lies in [table[b], table[b + 1]], so it is table[b] + the number of entries <= v among the next
    // W = widest bucket + 1 (entries past the answer are > v: the array is sorted and padded).  No data-dependent branch: what
    // costs in this loop is not arithmetic but mispredicted exits of a scan.
    cumsum(p, cdf0.data());
    int64_t W = 1;
    {
        const double total = cdf0[(size_t)n - 1];
        for (int64_t i = 0; i < n; ++i) cdf0[(size_t)i] = cdf0[(size_t)i] / total;
        int64_t i = 0;
        for (int64_t b = 0; b < LUT; ++b) {
            const double edge = (double)b / (double)LUT;
            while (cdf0[(size_t)i] <= edge && i < n - 1) ++i;
            lut0[(size_t)b] = (int32_t)i;
        }
        lut0[(size_t)LUT] = (int32_t)(n - 1);
        for (int64_t b = 0; b < LUT; ++b) W = std::max<int64_t>(W, lut0[(size_t)b + 1] - lut0[(size_t)b] + 1);
        W = (W + 3) / 4 * 4;
    }
    // searchsorted(c, v, "right") = number of entries <= v of a non-decreasing array, without data-dependent branches
    auto upper_bound = [&](const double *c, double v) {
        int64_t lo = 0, len = n;
        while (len > 1) {
            const int64_t half = len >> 1;
            lo += (c[lo + half - 1] <= v) ? half : 0;
            len -= half;
        }
        return lo + ((c[lo] <= v) ? 1 : 0);
    };
    constexpr int64_t LAZY = 24;                                 // rounds with fewer draws than this do not normalise the whole cdf
    int64_t pos = 0, done = 0, used = 0, round_id = 0;
    for (int64_t c = 0; c < cells; ++c) {
        int64_t *found = out + c * size;
        int64_t n_uniq = 0, zeroed = 0;
        bool fits = true;
        while (n_uniq < size) {
            const int64_t need = size - n_uniq;
            if (pos + need > pool_len) { fits = false; break; }
            const double *x = pool + pos;
            pos += need;
            ++round_id;
            int64_t added = 0;
            auto take = [&](int64_t lo) {                        // first occurrences, in draw order (np.unique(return_index) + sort + take)
                const int64_t fresh = stamp[(size_t)lo] != round_id;
                stamp[(size_t)lo] = round_id;
                found[n_uniq + added] = lo;                      // written either way, kept only if fresh: draw k of a round writes slot
                added += fresh;                                  // n_uniq + added <= n_uniq + k < size, and a later fresh index overwrites a repeat
            };
            if (n_uniq == 0 && W <= WMAX) {
                const double *cd = cdf0.data();
                for (int64_t k = 0; k < need; ++k) {
                    const double v = x[k];
                    const double *win = cd + lut0[(size_t)(int64_t)(v * (double)LUT)];
                    int64_t cnt = 0;
                    for (int64_t j = 0; j < W; ++j) cnt += win[j] <= v;
                    take((win - cd) + cnt);
                }
            } else if (n_uniq == 0) {
                for (int64_t k = 0; k < need; ++k) {
                    const int64_t lo = upper_bound(cdf0.data(), x[k]);
                    take(lo < n - 1 ? lo : n - 1);
                }
            } else {
                if (zeroed == 0) for (int64_t i = 0; i < n; ++i) pw[(size_t)i] = p[i];
                for (; zeroed < n_uniq; ++zeroed) pw[(size_t)found[zeroed]] = 0.0;
                cumsum(pw.data(), cs.data());
                const double total = cs[(size_t)n - 1];
                if (need >= LAZY) {
                    for (int64_t i = 0; i < n; ++i) cdf[(size_t)i] = cs[(size_t)i] / total;
                    for (int64_t k = 0; k < need; ++k) {
                        const int64_t lo = upper_bound(cdf.data(), x[k]);
                        take(lo < n - 1 ? lo : n - 1);
                    }
                } else {
                    // a handful of draws: look each one up in the UNNORMALISED sums near v * total, then settle the position with the
                    // exact predicate cs[i] / total <= v (the quotient is monotone in cs[i], so the normalised cdf is sorted the same
                    // way and its searchsorted is the first index where the predicate fails)
                    for (int64_t k = 0; k < need; ++k) {
                        const double v = x[k];
                        int64_t lo = upper_bound(cs.data(), v * total);
                        if (lo > n - 1) lo = n - 1;
                        while (lo > 0 && !(cs[(size_t)lo - 1] / total <= v)) --lo;
                        while (lo < n - 1 && cs[(size_t)lo] / total <= v) ++lo;
                        take(lo);
                    }
                }
            }
            n_uniq += added;
        }
        if (!fits) break;
        ++done;
        used = pos;
    }
    *cells_done = done;
    *consumed = used;
    return VCY_OK;
}
