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
    std::vector<double> pw((size_t)n), cdf((size_t)n), cdf0((size_t)n);
    std::vector<int64_t> stamp((size_t)n, -1);
    constexpr int64_t LUT = 128;                                 // power of two: v * LUT and b / LUT are exact
    std::vector<int32_t> lut((size_t)LUT), lut0((size_t)LUT);
    // cdf = cumsum(w) / cumsum(w)[-1] exactly as numpy forms it (sequential fp64 accumulation, elementwise division), plus a
    // bucket table over [0, 1): table[b] = first index with cdf > b / LUT, a lower bound of searchsorted(cdf, v, "right") for
    // every v in bucket b = floor(v * LUT); the few remaining steps are a scan.
    auto build = [&](const double *w, double *cd, int32_t *table) {
        double acc = 0.0;
        for (int64_t i = 0; i < n; ++i) { acc = acc + w[i]; cd[i] = acc; }
        const double total = cd[n - 1];
        for (int64_t i = 0; i < n; ++i) cd[i] = cd[i] / total;
        int64_t i = 0;
        for (int64_t b = 0; b < LUT; ++b) {
            const double edge = (double)b * (1.0 / LUT);
            while (cd[i] <= edge && i < n - 1) ++i;
            table[b] = (int32_t)i;
        }
    };
    build(p, cdf0.data(), lut0.data());                           // the first round of every cell sees the untouched p
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
            const double *cd = cdf0.data();
            const int32_t *table = lut0.data();
            if (n_uniq > 0) {
                if (zeroed == 0) for (int64_t i = 0; i < n; ++i) pw[(size_t)i] = p[i];
                for (; zeroed < n_uniq; ++zeroed) pw[(size_t)found[zeroed]] = 0.0;
                build(pw.data(), cdf.data(), lut.data());
                cd = cdf.data();
                table = lut.data();
            }
            ++round_id;
            int64_t added = 0;
            for (int64_t k = 0; k < need; ++k) {
                const double v = x[k];
                int64_t lo = table[(int64_t)(v * (double)LUT)];
                while (lo < n - 1 && cd[lo] <= v) ++lo;              // cdf[n-1] == 1 > v: numpy cannot run past the end either
                if (stamp[(size_t)lo] != round_id) { stamp[(size_t)lo] = round_id; found[n_uniq + added++] = lo; }
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
