/*
 * velocyto_oracle.c -- CPU restatement (fp64) of the velocyto.py analysis hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / the timed CPU baseline.  The product path (velocyto.py_amd/) never
 * calls into it and fails loudly when its HIP library is missing.
 *
 * Parity pin: the reference ships NO tests or golden vectors, so this
 * restatement is pinned against outputs of the reference itself run in the build
 * container: oracle/_ref (the reference's speedboosted.pyx compiled with its own flags)
 * and the reference's Python modules imported from /root/reference.  The pinned vectors
 * are committed under tests/golden/ together with tests/golden/make_golden.py.
 *
 * All matrices use the REFERENCE layout: (rows = genes, cols = cells), row-major fp64.
 * Each function cites the reference lines it restates (paths relative to
 * /root/reference/velocyto/).
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp; no -ffast-math: the oracle keeps strict
 * IEEE evaluation, the reference's -ffast-math reassociation is a tolerance item).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { VO_LINEAR = 0, VO_SQRT = 1, VO_LOG10 = 2 };

int vo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Element transform of t = e[g,i] - e[g,c].
 * full variants   : speedboosted.pyx:29 (linear), :110-114 (sqrt: t>0 ? sqrt(t+psc) : -sqrt(-t+psc)),
 *                   :195-199 (log10: same branch shape as sqrt-full)
 * partial variants: :282 (linear), :372-378 (sqrt: |t|<1e-16 -> 0, t>0 -> sqrt(t+psc), else -sqrt(-t+psc)),
 *                   :469-473 (log10: t>=0 ? log10(t+psc) : -log10(-t+psc))                         */
static inline double vo_transform(double t, int transform, int partial, double psc)
{
    switch (transform) {
    case VO_SQRT:
        if (partial && fabs(t) < 1e-16) return 0.0;
        return t > 0 ? sqrt(t + psc) : -sqrt(-t + psc);
    case VO_LOG10:
        if (partial) return t >= 0 ? log10(t + psc) : -log10(-t + psc);
        return t > 0 ? log10(t + psc) : -log10(-t + psc);
    default:
        return t;
    }
}

/* Pearson correlation over genes between column `i` of f(e - e[:,c]) and column `c` of d,
 * evaluated the way every x_colDeltaCor* body does it (speedboosted.pyx:31-78): mean of A,
 * centre, mean of b, centre, reciprocal root sums of squares, then the scaled dot product
 * accumulated gene by gene.  `a` is a caller-provided scratch of `rows` doubles.           */
static double vo_pair_corr(const double *e, const double *d, int rows, int cols, int c, int i,
                           int transform, int partial, double psc, double *a,
                           const double *b_centred, double inv_ssb)
{
    double mu = 0.0;
    for (int g = 0; g < rows; ++g) {
        double t = e[(size_t)g * cols + i] - e[(size_t)g * cols + c];
        a[g] = vo_transform(t, transform, partial, psc);
        mu += a[g];
    }
    mu /= rows;
    double ss = 0.0;
    for (int g = 0; g < rows; ++g) {
        a[g] -= mu;
        ss += a[g] * a[g];
    }
    double inv_ssa = 1.0 / sqrt(ss);          /* 1/sqrt(0) = inf  ->  0*inf = NaN, as in the reference */
    double r = 0.0;
    for (int g = 0; g < rows; ++g)
        r += (a[g] * inv_ssa) * (b_centred[g] * inv_ssb);
    (void)d;
    return r;
}

static void vo_centre_b(const double *d, int rows, int cols, int c, double *b, double *inv_ssb)
{
    double mu = 0.0;
    for (int g = 0; g < rows; ++g) mu += d[(size_t)g * cols + c];
    mu /= rows;
    double ss = 0.0;
    for (int g = 0; g < rows; ++g) {
        b[g] = d[(size_t)g * cols + c] - mu;
        ss += b[g] * b[g];
    }
    *inv_ssb = 1.0 / sqrt(ss);
}

/* Full variants: rm[c,i] += corr for every i.   speedboosted.pyx:13-87, 93-172, 178-257.  */
void vo_coldeltacor(const double *e, const double *d, double *rm, int rows, int cols,
                    int transform, double psc, int num_threads)
{
#ifdef _OPENMP
    if (num_threads > 0) omp_set_num_threads(num_threads);
#endif
#pragma omp parallel
    {
        double *a = (double *)malloc(sizeof(double) * rows);
        double *b = (double *)malloc(sizeof(double) * rows);
#pragma omp for schedule(dynamic, 1)
        for (int c = 0; c < cols; ++c) {
            double inv_ssb;
            vo_centre_b(d, rows, cols, c, b, &inv_ssb);
            for (int i = 0; i < cols; ++i)
                rm[(size_t)c * cols + i] += vo_pair_corr(e, d, rows, cols, c, i, transform, 0, psc, a, b, inv_ssb);
        }
        free(a);
        free(b);
    }
}

/* One cell of a partial variant, in the reference's own loop order (speedboosted.pyx:366-440): genes outer, neighbours inner - each gene
 * row of e is visited once and the nrndm neighbour values are gathered from inside that one row - over a per-thread scratch A of
 * rows x nrndm doubles: fill A, column means, centre, column sums of squares, centred b, then the scaled products accumulated gene by
 * gene.  (The reference keeps A and A - mean as two arrays; one array centred in place holds the same numbers.)  Per pair the sums run
 * over the genes in ascending order, exactly as in vo_pair_corr.  acc: nrndm doubles, zeroed here; b: rows doubles of scratch.          */
static void vo_partial_cell(const double *e, const double *d, int rows, int cols, int c, const int64_t *ix, int nrndm,
                            int transform, double psc, double *A, double *mu, double *b, double *acc)
{
    for (int n = 0; n < nrndm; ++n) { mu[n] = 0.0; acc[n] = 0.0; }
    for (int g = 0; g < rows; ++g) {
        const double *row = e + (size_t)g * cols;
        const double ec = row[c];
        double *Ag = A + (size_t)g * nrndm;
        for (int n = 0; n < nrndm; ++n) {
            Ag[n] = vo_transform(row[ix[n]] - ec, transform, 1, psc);
            mu[n] += Ag[n];
        }
    }
    for (int n = 0; n < nrndm; ++n) mu[n] /= rows;
    double inv_ssb;
    vo_centre_b(d, rows, cols, c, b, &inv_ssb);
    /* ss in acc for a moment */
    for (int g = 0; g < rows; ++g) {
        double *Ag = A + (size_t)g * nrndm;
        for (int n = 0; n < nrndm; ++n) {
            Ag[n] -= mu[n];
            acc[n] += Ag[n] * Ag[n];
        }
    }
    for (int n = 0; n < nrndm; ++n) { mu[n] = 1.0 / sqrt(acc[n]); acc[n] = 0.0; }   /* 1/sqrt(0) = inf -> 0 * inf = NaN, as in the reference */
    for (int g = 0; g < rows; ++g) {
        const double *Ag = A + (size_t)g * nrndm;
        const double bg = b[g] * inv_ssb;
        for (int n = 0; n < nrndm; ++n) acc[n] += (Ag[n] * mu[n]) * bg;
    }
}

/* Partial variants: only i = ixs[c,n]; scatter-add into the dense (cols,cols) matrix.
 * speedboosted.pyx:263-346, 352-443, 449-538.                                          */
void vo_coldeltacor_partial(const double *e, const double *d, double *rm, const int64_t *ixs,
                            int rows, int cols, int nrndm, int transform, double psc, int num_threads)
{
#ifdef _OPENMP
    if (num_threads > 0) omp_set_num_threads(num_threads);
#endif
#pragma omp parallel
    {
        double *A = (double *)malloc(sizeof(double) * (size_t)rows * nrndm);
        double *b = (double *)malloc(sizeof(double) * rows);
        double *mu = (double *)malloc(sizeof(double) * 2 * nrndm), *acc = mu + nrndm;
#pragma omp for schedule(dynamic, 1)
        for (int c = 0; c < cols; ++c) {
            const int64_t *ix = ixs + (size_t)c * nrndm;
            vo_partial_cell(e, d, rows, cols, c, ix, nrndm, transform, psc, A, mu, b, acc);
            for (int n = 0; n < nrndm; ++n) rm[(size_t)c * cols + ix[n]] += acc[n];
        }
        free(A);
        free(b);
        free(mu);
    }
}

/* Same as vo_coldeltacor_partial but with the compact (cols, nrndm) result the GPU path
 * produces natively (out[c,n] = corr with neighbour ixs[c,n]); used for at-scale checks
 * where a dense (cols,cols) fp64 matrix is not affordable, and by the CPU baseline.
 * Cells c in [c0, c1) only: e and d keep their full width, so a cell range of a large problem
 * is read with the large problem's strides.                                                */
void vo_coldeltacor_partial_compact(const double *e, const double *d, double *out, const int64_t *ixs,
                                    int rows, int cols, int nrndm, int transform, double psc,
                                    int c0, int c1, int num_threads)
{
#ifdef _OPENMP
    if (num_threads > 0) omp_set_num_threads(num_threads);
#endif
#pragma omp parallel
    {
        double *A = (double *)malloc(sizeof(double) * (size_t)rows * nrndm);
        double *b = (double *)malloc(sizeof(double) * rows);
        double *mu = (double *)malloc(sizeof(double) * 2 * nrndm), *acc = mu + nrndm;
#pragma omp for schedule(dynamic, 1)
        for (int c = c0; c < c1; ++c) {
            vo_partial_cell(e, d, rows, cols, c, ixs + (size_t)c * nrndm, nrndm, transform, psc, A, mu, b, acc);
            for (int n = 0; n < nrndm; ++n) out[(size_t)c * nrndm + n] = acc[n];
        }
        free(A);
        free(b);
        free(mu);
    }
}

/* convolve_by_sparse_weights (neighbors.py:416-423): out = data @ w.T with w a (cols,cols)
 * CSR weight matrix, i.e. out[g,c] = sum_j w[c,j] * data[g,j].  `out` is (rows, cols)
 * row-major here (the reference returns the same values Fortran-ordered).               */
void vo_convolve_csr(const double *data, const int64_t *indptr, const int64_t *indices,
                     const double *w, double *out, int rows, int cols, int num_threads)
{
#ifdef _OPENMP
    if (num_threads > 0) omp_set_num_threads(num_threads);
#endif
#pragma omp parallel for schedule(static)
    for (int g = 0; g < rows; ++g) {
        const double *row = data + (size_t)g * cols;
        for (int c = 0; c < cols; ++c) {
            double acc = 0.0;
            for (int64_t p = indptr[c]; p < indptr[c + 1]; ++p)
                acc += w[p] * row[indices[p]];
            out[(size_t)g * cols + c] = acc;
        }
    }
}

/* fit_slope / _fit1_slope (estimation.py:173-188, 267-279): per gene, scipy.optimize.nnls on a
 * single column x against y  ==  max(0, <x,y>/<x,x>);  all-zero x -> NaN, all-zero y -> 0.
 * Output is float32 like the reference (`dtype="float32"`, :277).                             */
void vo_fit_slope(const double *Y, const double *X, float *slopes, int rows, int cols)
{
    for (int g = 0; g < rows; ++g) {
        const double *x = X + (size_t)g * cols, *y = Y + (size_t)g * cols;
        int anyx = 0, anyy = 0;
        double sxy = 0.0, sxx = 0.0;
        for (int c = 0; c < cols; ++c) {
            anyx |= (x[c] != 0.0);
            anyy |= (y[c] != 0.0);
            sxy += x[c] * y[c];
            sxx += x[c] * x[c];
        }
        double m;
        if (!anyx) m = NAN;
        else if (!anyy) m = 0.0;
        else {
            m = sxy / sxx;
            if (m < 0.0) m = 0.0;
        }
        slopes[g] = (float)m;
    }
}

/* balance_knn_loop / balance_knn_loop_constrained (neighbors.py:11-72, 75-140).
 * dsi (n, K) distance-sorted sight indices, dist (n, K), lsi processing order,
 * groups == NULL for the unconstrained loop.  Outputs dist_new (n,k+1), dsi_new (n,k+1)
 * initialised to -1 (:43), l (n).  The pad-with-self tail (:65-69) fires only when the sight
 * loop ran to its last column without a `break` and p < k; dist_new is written there even when
 * return_distance is False in the reference (then replaced by ones, :70-71).               */
void vo_balance_knn(const int64_t *dsi, const double *dist, const int64_t *lsi, const int64_t *groups,
                    int64_t n, int64_t K, int64_t maxl, int64_t k, int return_distance,
                    double *dist_new, int64_t *dsi_new, int64_t *l)
{
    for (int64_t t = 0; t < n * (k + 1); ++t) { dsi_new[t] = -1; dist_new[t] = 0.0; }
    memset(l, 0, sizeof(int64_t) * n);
    for (int64_t i = 0; i < n; ++i) {
        int64_t el = lsi[i], p = 0, j = 0, last_j = 0;
        for (j = 0; j < K; ++j) {
            last_j = j;
            if (p >= k) break;
            int64_t m = dsi[el * K + j];
            if (el == m) { dsi_new[el * (k + 1)] = el; continue; }
            if (groups && groups[el] != groups[m]) continue;
            if (l[m] >= maxl) continue;
            dsi_new[el * (k + 1) + p + 1] = m;
            l[m] += 1;
            if (return_distance) dist_new[el * (k + 1) + p + 1] = dist[el * K + j];
            p += 1;
        }
        if (last_j == K - 1 && p < k) {
            while (p < k) {
                dsi_new[el * (k + 1) + p + 1] = el;
                dist_new[el * (k + 1) + p + 1] = dist[el * K];
                p += 1;
            }
        }
    }
    if (!return_distance)
        for (int64_t t = 0; t < n * (k + 1); ++t) dist_new[t] = 1.0;
}
