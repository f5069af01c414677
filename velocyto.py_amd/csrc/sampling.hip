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

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace {
constexpr int64_t WMAX = 32;                                     // widest bucket the fixed-length scan of round 1 is used for
constexpr int64_t LAZY = 24;                                     // rounds with fewer draws than this do not normalise the whole cdf

// what every cell shares: p and the first round's cdf with its bucket table
struct ChoiceTable {
    const double *p;
    int64_t n, size, LUT, W;
    std::vector<double> cdf0;
    std::vector<int32_t> lut0;
};
// what a replaying thread owns
struct ChoiceScratch {
    std::vector<double> pw, cs, cdf;
    std::vector<int64_t> stamp;
    int64_t round_id = 0;
    explicit ChoiceScratch(int64_t n) : pw((size_t)n), cs((size_t)n), cdf((size_t)n), stamp((size_t)n, -1) {}
};

// cdf = cumsum(w) / cumsum(w)[-1] exactly as numpy forms it: a sequential fp64 accumulation, then an elementwise division.
inline void cumsum(const double *w, double *c, int64_t n)
{
#pragma clang fp contract(off)
    double acc = 0.0;
    for (int64_t i = 0; i < n; ++i) { acc = acc + w[i]; c[i] = acc; }
}
// searchsorted(c, v, "right") = number of entries <= v of a non-decreasing array, without data-dependent branches
inline int64_t upper_bound(const double *c, int64_t n, double v)
{
    int64_t lo = 0, len = n;
    while (len > 1) {
        const int64_t half = len >> 1;
        lo += (c[lo + half - 1] <= v) ? half : 0;
        len -= half;
    }
    return lo + ((c[lo] <= v) ? 1 : 0);
}

void build_table(ChoiceTable &t, const double *p, int64_t n, int64_t size)
{
#pragma clang fp contract(off)
    t.p = p; t.n = n; t.size = size;
    t.LUT = 128;                                                 // power of two: v * LUT and b / LUT are exact; about two buckets
    while (t.LUT < 2 * n && t.LUT < 4096) t.LUT *= 2;            // per entry, so that the scan below is a few entries long
    t.cdf0.assign((size_t)(n + WMAX), 2.0);                      // padded with values no draw reaches
    t.lut0.assign((size_t)t.LUT + 1, 0);
    // The first round of every cell sees the untouched p and takes most of the draws: its cdf gets a bucket table over [0, 1),
    // table[b] = first index with cdf > b / LUT, a lower bound of searchsorted(cdf, v, "right") for every v in bucket
    // b = floor(v * LUT); the answer lies in [table[b], table[b + 1]], so it is table[b] + the number of entries <= v among the next
    // W = widest bucket + 1 (entries past the answer are > v: the array is sorted and padded).  No data-dependent branch: what
    // costs in this loop is not arithmetic but mispredicted exits of a scan.
    cumsum(p, t.cdf0.data(), n);
    const double total = t.cdf0[(size_t)n - 1];
    for (int64_t i = 0; i < n; ++i) t.cdf0[(size_t)i] = t.cdf0[(size_t)i] / total;
    int64_t i = 0;
    for (int64_t b = 0; b < t.LUT; ++b) {
        const double edge = (double)b / (double)t.LUT;
        while (t.cdf0[(size_t)i] <= edge && i < n - 1) ++i;
        t.lut0[(size_t)b] = (int32_t)i;
    }
    t.lut0[(size_t)t.LUT] = (int32_t)(n - 1);
    t.W = 1;
    for (int64_t b = 0; b < t.LUT; ++b) t.W = std::max<int64_t>(t.W, t.lut0[(size_t)b + 1] - t.lut0[(size_t)b] + 1);
    t.W = (t.W + 3) / 4 * 4;
}

// One np.random.choice(n, size, replace=False, p=p) over the uniforms pool[pos...]: the chosen indices in `found` (size entries),
// returns the position after the uniforms it took, or -1 when the pool ends first (found is then unspecified).
int64_t replay_cell(const ChoiceTable &t, ChoiceScratch &s, const double *pool, int64_t pool_len, int64_t pos, int64_t *found)
{
#pragma clang fp contract(off)
    const int64_t n = t.n, size = t.size;
    int64_t n_uniq = 0, zeroed = 0;
    while (n_uniq < size) {
        const int64_t need = size - n_uniq;
        if (pos + need > pool_len) return -1;
        const double *x = pool + pos;
        pos += need;
        const int64_t round_id = ++s.round_id;
        int64_t added = 0;
        int64_t *stamp = s.stamp.data();
        auto take = [&](int64_t lo) {                            // first occurrences, in draw order (np.unique(return_index) + sort + take)
            const int64_t fresh = stamp[lo] != round_id;
            stamp[lo] = round_id;
            found[n_uniq + added] = lo;                          // written either way, kept only if fresh: draw k of a round writes slot
            added += fresh;                                      // n_uniq + added <= n_uniq + k < size, and a later fresh index overwrites a repeat
        };
        if (n_uniq == 0 && t.W <= WMAX) {
            const double *cd = t.cdf0.data();
            const int32_t *lut = t.lut0.data();
            const double scale = (double)t.LUT;
            const int64_t W = t.W;
            for (int64_t k = 0; k < need; ++k) {
                const double v = x[k];
                const double *win = cd + lut[(int64_t)(v * scale)];
                int64_t cnt = 0;
                for (int64_t j = 0; j < W; ++j) cnt += win[j] <= v;
                take((win - cd) + cnt);
            }
        } else if (n_uniq == 0) {
            for (int64_t k = 0; k < need; ++k) {
                const int64_t lo = upper_bound(t.cdf0.data(), n, x[k]);
                take(lo < n - 1 ? lo : n - 1);
            }
        } else {
            double *pw = s.pw.data(), *cs = s.cs.data();
            if (zeroed == 0) std::memcpy(pw, t.p, (size_t)n * sizeof(double));
            for (; zeroed < n_uniq; ++zeroed) pw[found[zeroed]] = 0.0;
            cumsum(pw, cs, n);
            const double total = cs[n - 1];
            if (need >= LAZY) {
                double *cdf = s.cdf.data();
                for (int64_t i = 0; i < n; ++i) cdf[i] = cs[i] / total;
                for (int64_t k = 0; k < need; ++k) {
                    const int64_t lo = upper_bound(cdf, n, x[k]);
                    take(lo < n - 1 ? lo : n - 1);
                }
            } else {
                // a handful of draws: look each one up in the UNNORMALISED sums near v * total, then settle the position with the
                // exact predicate cs[i] / total <= v (the quotient is monotone in cs[i], so the normalised cdf is sorted the same
                // way and its searchsorted is the first index where the predicate fails)
                for (int64_t k = 0; k < need; ++k) {
                    const double v = x[k];
                    int64_t lo = upper_bound(cs, n, v * total);
                    if (lo > n - 1) lo = n - 1;
                    while (lo > 0 && !(cs[lo - 1] / total <= v)) --lo;
                    while (lo < n - 1 && cs[lo] / total <= v) ++lo;
                    take(lo);
                }
            }
        }
        n_uniq += added;
    }
    return pos;
}

// A chain of cells replayed from a GUESSED pool position (a worker's share of the stream): start[k] = where its k-th cell began
// (start[count] = where the chain stopped), rows = the indices it chose.
struct Chain {
    std::vector<int64_t> start, rows;
    int64_t count = 0;
    bool pool_ended = false;
};

void run_chain(const ChoiceTable &t, const double *pool, int64_t pool_len, int64_t pos, int64_t max_cells, Chain &ch)
{
    ChoiceScratch s(t.n);
    ch.start.resize((size_t)max_cells + 1);
    ch.rows.resize((size_t)(max_cells * t.size));
    ch.count = 0;
    ch.start[0] = pos;
    while (ch.count < max_cells) {
        const int64_t next = replay_cell(t, s, pool, pool_len, pos, ch.rows.data() + ch.count * t.size);
        if (next < 0) { ch.pool_ended = true; break; }
        pos = next;
        ch.start[(size_t)++ch.count] = pos;
    }
}

int64_t env_or(const char *name, int64_t dflt)
{
    const char *e = std::getenv(name);
    return e && *e ? std::atoll(e) : dflt;
}
int choice_threads()                                             // VCY_CHOICE_THREADS (default 8, at most the hardware's)
{
    int t = (int)env_or("VCY_CHOICE_THREADS", 8);
    const int hw = (int)std::thread::hardware_concurrency();
    if (hw > 0 && t > hw) t = hw;
    return t < 1 ? 1 : t;
}
}  // namespace

// pool: `pool_len` uniforms in [0, 1) in stream order.  p: n probabilities (already normalised as the caller passes them to
// numpy).  out: (cells, size) int64.  Returns VCY_OK and sets *cells_done (cells whose draws fitted in the pool) and *consumed
// (uniforms those cells took: the caller advances its RandomState by exactly that many).
//
// The stream is sequential - where a cell's uniforms begin is only known when the cell before it is done - but not hopelessly so:
// a replay started at ANY position of the pool soon runs in step with the true one, because both advance by a cell's take (a few
// dozen distinct values) and are identical from the first position they share.  So the cells are split among worker threads, each
// replaying a chain from a guessed position (cells before it x the mean take measured on a sequential prefix); the true chain is
// then stitched together: it follows worker s until it lands on a position worker s + 1 visited, and continues there.  Same
// results as the sequential replay by construction; if two chains have not met within the slack, the true one is simply replayed
// on (sequentially) until they do.  (Measured at n = 501, size = 250 - 315 uniforms per cell: chains meet after ~300 cells.)
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
    ChoiceTable t;
    build_table(t, p, n, size);
    ChoiceScratch s0(n);
    int64_t pos = 0, done = 0;
    auto sequential = [&](int64_t upto) {                        // the true chain, cell by cell; false when the pool ended
        while (done < upto) {
            const int64_t next = replay_cell(t, s0, pool, pool_len, pos, out + done * size);
            if (next < 0) return false;
            pos = next;
            ++done;
        }
        return true;
    };
    const int T = choice_threads();
    // (the three knobs below exist for the tests: tiny shares and no slack drive the stitching through its rare branches)
    const int64_t PREFIX = std::max<int64_t>(1, env_or("VCY_CHOICE_PREFIX", 128)), MIN_PER_THREAD = std::max<int64_t>(1, env_or("VCY_CHOICE_MIN_SHARE", 512));
    bool alive = sequential(std::min(cells, PREFIX));
    const int64_t rest = cells - done;
    int workers = (int)std::min<int64_t>(T, rest / MIN_PER_THREAD);
    if (alive && size > 0 && workers >= 2) {
        const double take = (double)pos / (double)done;          // uniforms per cell so far
        const int64_t per = (rest + workers - 1) / workers;
        // cells a worker replays past its share.  A chain lands on a position of the next one with a chance of about 1 / take per
        // cell (that chain visits one position in `take`), so they meet after about `take` cells; twice that leaves one handover
        // in seven to the sequential continuation below, which costs the same per cell but runs alone.
        const int64_t slack = std::max<int64_t>(0, env_or("VCY_CHOICE_SLACK", 64 + per / 32 + (int64_t)(2.0 * take)));
        std::vector<Chain> chains((size_t)workers);
        std::vector<std::thread> pool_threads;
        for (int w = 0; w < workers; ++w) {
            const int64_t guess = w == 0 ? pos : std::min(pool_len, pos + (int64_t)((double)(w * per) * take));
            pool_threads.emplace_back(run_chain, std::cref(t), pool, pool_len, guess, per + slack, std::ref(chains[(size_t)w]));
        }
        for (auto &th : pool_threads) th.join();
        // stitch: follow chain w from its cell k; hand over to chain w + 1 at the first position both visited
        int w = 0;
        int64_t k = 0;
        while (alive && done < cells) {
            Chain &a = chains[(size_t)w];
            int64_t stop = a.count, jn = -1;                     // cells [k, stop) of chain w are taken; jn = where chain w + 1 is entered
            if (w + 1 < workers) {
                const Chain &b = chains[(size_t)w + 1];
                int64_t i = k, j = 0;
                while (i <= a.count && j <= b.count) {           // both start lists ascend: merge until a common position
                    if (a.start[(size_t)i] == b.start[(size_t)j]) { stop = i; jn = j; break; }
                    if (a.start[(size_t)i] < b.start[(size_t)j]) ++i; else ++j;
                }
            }
            const int64_t m = std::min(stop - k, cells - done);
            std::memcpy(out + done * size, a.rows.data() + k * size, (size_t)(m * size) * sizeof(int64_t));
            done += m;
            pos = a.start[(size_t)(k + m)];
            if (done == cells) break;
            if (jn >= 0) { ++w; k = jn; continue; }
            // chain w ended without meeting the next one (or it is the last): go on cell by cell from where it stopped
            if (a.pool_ended && stop == a.count) { alive = false; break; }
            if (w + 1 < workers) {
                const Chain &b = chains[(size_t)w + 1];
                bool met = false;
                while (done < cells) {
                    const auto it = std::lower_bound(b.start.begin(), b.start.begin() + b.count + 1, pos);
                    if (it != b.start.begin() + b.count + 1 && *it == pos) { k = it - b.start.begin(); ++w; met = true; break; }
                    if (!sequential(done + 1)) { alive = false; break; }
                }
                if (!met) break;
            } else {
                alive = sequential(cells);
                break;
            }
        }
    } else if (alive) {
        alive = sequential(cells);
    }
    *cells_done = done;
    *consumed = pos;
    return VCY_OK;
}

