/*
 * velocyto_hip.h -- C ABI of libvelocyto_hip.so, the MI355X (gfx950) implementation of the
 * velocyto.py post-counting analysis hot path.
 *
 * The reference has no FFI for this path: its "operator API" is a set of Python call
 * surfaces.  Each entry point below names the reference interface it
 * stands behind (paths relative to /root/reference/velocyto/).  A maintainer binds them
 * with ctypes exactly as velocyto.py_amd/_lib.py does; see INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 (VCY_OK) or a negative status; vcy_last_error() returns a
 *     thread-local message for the last failure.
 *   - all array arguments are DEVICE pointers unless the name ends in `_host`.
 *   - matrices are CELLS-MAJOR: shape (C cells, G genes), row-major with leading dimension
 *     `ld` (elements, >= G, multiple of 4 for f32 / 2 for f64 so rows stay 16-byte aligned).
 *     This is the reference's Fortran-ordered (G,C) array (what convolve_by_sparse_weights
 *     returns) -- one cell's gene vector is one contiguous stream.
 *   - `dtype` selects the storage/compute type of the matrices: VCY_F32 (production) or
 *     VCY_F64 (reference precision; used by the parity tests).
 *   - asynchronous on `stream` (a hipStream_t), no hidden synchronisation, no allocation:
 *     outputs and workspaces are caller-provided (vcy_*_workspace_bytes report sizes).
 *   - the caller owns every buffer; inputs are never written.
 */
#ifndef VELOCYTO_HIP_H
#define VELOCYTO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VCY_OK 0
#define VCY_ERR_INVALID (-1)      /* bad argument (shape, alignment, enum)          */
#define VCY_ERR_HIP (-2)          /* HIP runtime error (launch failure, no device)  */
#define VCY_ERR_UNSUPPORTED (-3)  /* valid request outside what the kernels cover   */

typedef enum { VCY_F32 = 0, VCY_F64 = 1, VCY_U16 = 2, VCY_U8 = 3 /* U16 / U8: count matrices (vcy_transpose, vcy_knn_pool_counts, vcy_gene_stats) */ } vcy_dtype;

/* Element transform f of the six reference kernels (speedboosted.pyx). */
typedef enum { VCY_LINEAR = 0, VCY_SQRT = 1, VCY_LOG10 = 2 } vcy_transform;

/* Which family's branch rules apply at t == 0 (speedboosted.pyx:110-114,195-199 vs
 * :372-378,469-473). */
typedef enum {
    VCY_RULES_FULL = 0,
    VCY_RULES_PARTIAL = 1,
    /* The partial rule with the pseudocount dropped: A = sign(t) sqrt|t| (0 at t == 0).  VCY_SQRT on VCY_F32 matrices in the
     * vcy_coldeltacor_partial* entries only (anything else: VCY_ERR_INVALID).  In f32 `|t| + psc` rounds to `|t|` for every
     * |t| >= 2^24 psc (1.7e-3 at the reference's default psc = 1e-10); below that the two rules differ by at most
     * psc / (2 sqrt|t|) per gene.  Three VALU instructions per gene instead of five: the caller opts in when psc is far
     * below the scale of the matrix (the Python layer: psc <= 1e-9, mean |e| >= 1e-4, no non-zero entry below 1e-20:
     * v_rsq_f32 reads denormal differences, |t| < 2^-126, as zero and the product would be infinite).                  */
    VCY_RULES_PARTIAL_NOPSC = 2
} vcy_rules;

typedef void *vcy_stream;  /* hipStream_t */

const char *vcy_last_error(void);
/* 4 (round 6).  History: 1 -> 2: vcy_diffuse_step_factored gained `int prepared` before compute_dtype; vcy_gram and vcy_embedding_scaling added;
 * vcy_knn_pool_csr requires >= 4 stored elements.  2 -> 3: vcy_clock_probe and vcy_coldeltacor_full_linear added.  3 -> 4: vcy_coldeltacor_full_linear_workspace_bytes takes C_out (repair flags); vcy_gemm_nt added.  A binder refuses a library whose version it was not built
 * against (velocyto_amd/_lib.py: EXPECTED_ABI). */
int vcy_abi_version(void);
/* Number of CUs / LDS bytes per workgroup of the current device (host query). */
int vcy_device_info(int *cu_count, int *lds_bytes_per_block, int64_t *hbm_bytes);
/* Measurement aid (no reference counterpart): the shader clock while other kernels run.  `nblocks` one-wave workgroups (workgroup b runs on
 * XCD b % 8) each take `nsamples` readings, `interval_ticks` ticks of the constant 100 MHz counter apart, of (shader-clock counter,
 * 100 MHz counter) into samples[b][i][2] (device int64).  The CUs run at the clock the power budget allows: between two readings
 * d(shader) / d(100 MHz) x 0.1 is the clock in GHz.  Launch it on a side stream before the kernel of interest; it ends by itself after
 * nsamples x interval_ticks (at most 3 s).                                                                                          */
int vcy_clock_probe(int64_t *samples, int64_t nblocks, int64_t nsamples, int64_t interval_ticks, vcy_stream stream);

/* ---------------------------------------------------------------- layout plumbing
 * (G,C) genes-major  <->  (C,ld) cells-major tiled transpose with optional dtype change.
 * Replaces the implicit layout of the reference's numpy arrays (analysis.py:59-61).
 * src is (rows, cols) row-major with leading dim ld_src; dst is (cols, rows) with ld_dst.
 * Padding columns of dst (rows <= j < ld_dst) are zero-filled.                          */
int vcy_transpose(const void *src, void *dst, int64_t rows, int64_t cols, int64_t ld_src, int64_t ld_dst,
                  int src_dtype, int dst_dtype, vcy_stream stream);

/* ---------------------------------------------------------------- stage D: correlations
 * speedboosted._colDeltaCorpartial / _colDeltaCorSqrtpartial / _colDeltaCorLog10partial
 * (speedboosted.pyx:263-538, 574-610; wrappers estimation.py:36-62, 90-116, 144-170).
 * For every cell c and every listed neighbour i = ixs[c,n]:
 *     out[c,n] = pearson_g( f(e[i,g] - e[c,g]), d[c,g] )
 * e: (C, ld) cells-major, all cells (any cell can be a neighbour).  d: cells-major rows for cells
 * d_row0, d_row0+1, ... (d_row0 = 0 for a full matrix; a cell-sharded rank passes only its own
 * rows with d_row0 = cell0).  ixs: (C_out, nrndm) int32 rows for cells cell0..cell0+C_out-1.
 * out : (C_out, nrndm) of `dtype` -- the COMPACT form of the reference's dense (C,C) `rm`
 * (use vcy_scatter_rows to materialise rm[c, ixs[c,n]] += out[c,n]).
 * Zero-variance columns give NaN exactly like the reference (0 * inf).
 * `order` (optional, may be NULL): C_out cell numbers (relative to cell0) in the order in which they are
 * scheduled (locality-sorted orders raise cache reuse of shared neighbours); results do not depend on it.
 * With `order`, rows of ixs / out / d are addressed by the cell numbers it holds, so C_out may be SMALLER than the
 * number of rows of ixs / out: only the listed cells are computed, the other rows of out are left untouched (a
 * cell-sharded rank computes the cells whose neighbours are all local while the halo exchange is in flight).
 * `rules` = VCY_RULES_PARTIAL reproduces the *partial kernels, VCY_RULES_FULL the branch
 * rules of the full kernels on an explicit neighbour list.                                */
int vcy_coldeltacor_partial(const void *e, const void *d, const int32_t *ixs, void *out, const int32_t *order,
                            int64_t C, int64_t G, int64_t ld, int64_t cell0, int64_t C_out, int64_t d_row0,
                            int64_t nrndm, int transform, int rules, double psc, int dtype, vcy_stream stream);

/* Stage C folded into stage D: the same correlations with d = the dmat of estimate_transition_prob computed on the fly from
 * the velocity chain (analysis.py:1346, 1369, 1399, 1538, 1575-1601; constant_velocity assumption),
 *     dmat = sign(D) f(|D| + psc),  D = (Sx + used_dt * dt_shift * (Ux - (gamma Sx + q))) - Sx,   f as `transform`,
 * so that neither velocity nor dmat is materialised.  e = Sx_sz (C, ld); Ux_sz holds rows u_row0..; gamma, q (G) float32
 * (q may be NULL).  Results are bit-identical to vcy_velocity_chain followed by vcy_coldeltacor_partial.             */
int vcy_coldeltacor_partial_fused(const void *Sx_sz, const void *Ux_sz, const float *gamma, const float *q, const int32_t *ixs,
                                  void *out, const int32_t *order, int64_t C, int64_t G, int64_t ld, int64_t cell0, int64_t C_out,
                                  int64_t u_row0, int64_t nrndm, int transform, int rules, double psc, double dt_shift,
                                  double used_dt, int dtype, vcy_stream stream);

/* The real and the randomised-control correlations of estimate_transition_prob(calculate_randomized=True), the
 * reference's default (analysis.py:1539-1542: delta_S_rndm = permute_rows_nsign(delta_S); :1578-1601: the second
 * colDeltaCor*partial call with dmat_rndm on the SAME e and neighbour lists), in ONE pass: A = f(e_i - e_c) is evaluated
 * once per pair and gene and correlated against both d[c] and d_rndm[c].  out_rndm has the shape of out; d_rndm the row
 * range of d.  Each output equals what vcy_coldeltacor_partial returns for that d alone: bit for bit on
 * VCY_F32 (same chunk length, hence the same order of summation), up to the rounding of the moment sums on VCY_F64 (the
 * f64 single kernel walks longer chunks; 1e-13 absolute on the correlations); problems too small for the grouped kernel
 * (< 8 neighbours or < 24 cells) run the single kernel twice.                                                           */
int vcy_coldeltacor_partial_dual(const void *e, const void *d, const void *d_rndm, const int32_t *ixs, void *out, void *out_rndm,
                                 const int32_t *order, int64_t C, int64_t G, int64_t ld, int64_t cell0, int64_t C_out, int64_t d_row0,
                                 int64_t nrndm, int transform, int rules, double psc, int dtype, vcy_stream stream);

/* vcy_coldeltacor_partial_fused + the randomised control in the same pass: d[c] from the velocity chain on the fly,
 * d_rndm (rows u_row0..) the materialised transform of the permuted delta_S (analysis.py:1540-1541, 1575-1601).        */
int vcy_coldeltacor_partial_fused_dual(const void *Sx_sz, const void *Ux_sz, const float *gamma, const float *q, const void *d_rndm,
                                       const int32_t *ixs, void *out, void *out_rndm, const int32_t *order, int64_t C, int64_t G,
                                       int64_t ld, int64_t cell0, int64_t C_out, int64_t u_row0, int64_t nrndm, int transform, int rules,
                                       double psc, double dt_shift, double used_dt, int dtype, vcy_stream stream);

/* speedboosted._colDeltaCor / _colDeltaCorSqrt / _colDeltaCorLog10 (speedboosted.pyx:13-257,
 * 542-572; wrappers estimation.py:11-33, 65-87, 119-141): all pairs.
 * rm is the dense (C_out, ld_rm) row block for cells cell0..cell0+C_out-1, columns 0..C-1;
 * accumulate != 0 reproduces the reference's `rm[c,i] += ...`, 0 overwrites.             */
int vcy_coldeltacor_full(const void *e, const void *d, void *rm, int64_t C, int64_t G, int64_t ld,
                         int64_t cell0, int64_t C_out, int64_t ld_rm, int transform, double psc,
                         int accumulate, int dtype, vcy_stream stream);

/* The LINEAR all-pairs variant, speedboosted._colDeltaCor (speedboosted.pyx:13-87; estimation.py:11-33), as the dense contraction it is:
 * sum A = Se_i - Se_c, sum A^2 = See_i + See_c - 2 (E E^T)[c][i], sum A b = (D E^T)[c][i] - sum_g e_c[g] d_c[g], the two products over the
 * genes on the f64 matrix cores (v_mfma_f64_16x16x4_f64, 128 x 64 tiles) with Pearson's r and the `rm[c][i] (+)= r` store fused into the
 * epilogue.  The expansion cancels where e_i is close to e_c (or d_c close to constant): pairs whose variance is below 2^-10 of the terms
 * it was expanded from are flagged by the epilogue and evaluated by a second, small launch in the reference's own form - A = e_i - e_c
 * element by element, centred, then squared (speedboosted.pyx:29-78) - so that near-duplicate cells carry the accuracy of every other pair
 * and NaN appears exactly where the reference's 0 * inf does (i == c, exact duplicates, a constant d_c).  Same arguments as
 * vcy_coldeltacor_full; workspace: vcy_coldeltacor_full_linear_workspace_bytes(C, C_out) bytes (five f64 sums per cell + one flag bit per
 * pair), 8-byte aligned.  f64 arithmetic whatever `dtype` the matrices are stored in.  ld must be a multiple of 16 and the columns
 * G .. ld - 1 of e and d zero (the cells-major layout's padding, as vcy_transpose writes it).                                             */
size_t vcy_coldeltacor_full_linear_workspace_bytes(int64_t C, int64_t C_out);
int vcy_coldeltacor_full_linear(const void *e, const void *d, void *rm, void *workspace, int64_t C, int64_t G, int64_t ld, int64_t cell0,
                                int64_t C_out, int64_t ld_rm, int accumulate, int dtype, vcy_stream stream);

/* rm[c, ixs[c,n]] += vals[c,n] (atomic: duplicate neighbours accumulate like the reference's
 * scatter at speedboosted.pyx:332-336).  rm: (C_out, ld_rm) of `dtype`.                    */
int vcy_scatter_rows(const void *vals, const int32_t *ixs, void *rm, int64_t C_out, int64_t nrndm,
                     int64_t ld_rm, int dtype, vcy_stream stream);

/* ---------------------------------------------------------------- stage A: kNN pooling
 * neighbors.convolve_by_sparse_weights (neighbors.py:416-423) as used by
 * VelocytoLoom.knn_imputation[_precomputed] (analysis.py:1011-1019, 1046-1050):
 *     out[c,:] = sum_p w[p] * data[indices[p],:]   for p in [indptr[c], indptr[c+1])
 * i.e. (data @ w.T) in the reference's layout, w a CSR (C_out x C) weight matrix whose rows
 * sum to one.  maximum != 0 additionally takes max(out[c,:], data[cell0+c,:]).
 * slab_genes: genes per pass (0 = default); the launch walks gene slabs so that one slab of
 * all cells (C * slab * 4 B) stays resident in the 256 MiB Infinity Cache while it is
 * gathered k times.  order (optional, NULL = natural): permutation of 0..C_out-1, the order in
 * which cells are scheduled within a slab; a locality-sorted order (cells close in the kNN space
 * next to each other) lets co-resident workgroups share neighbour rows in the per-XCD L2.
 * Results do not depend on it.                                                              */
int vcy_knn_pool(const void *data, void *out, const int64_t *indptr, const int32_t *indices, const void *w,
                 const int32_t *order, int64_t C, int64_t G, int64_t ld, int64_t cell0, int64_t C_out, int maximum,
                 int64_t slab_genes, int dtype, vcy_stream stream);

/* Same, pooling TWO matrices that share the weights in one launch (spliced and unspliced,
 * analysis.py:1012-1013): indices/weights are read once and twice as many gathers are in flight.  */
int vcy_knn_pool2(const void *data, void *out, const void *data2, void *out2, const int64_t *indptr,
                  const int32_t *indices, const void *w, const int32_t *order, int64_t C, int64_t G, int64_t ld,
                  int64_t cell0, int64_t C_out, int maximum, int64_t slab_genes, int dtype, vcy_stream stream);

/* ONE matrix pooled with TWO weight sets over the same graph, out = data-rows . w, out2 = data-rows . w2: the two
 * `hi_dim @ (transition_prob - embedding_knn / n).T` products of calculate_embedding_shift (analysis.py:1716 real, :1728
 * randomised control) share hi_dim and the neighbour lists, so the rows are gathered once.                      */
int vcy_knn_pool_w2(const void *data, void *out, void *out2, const int64_t *indptr, const int32_t *indices, const void *w,
                    const void *w2, const int32_t *order, int64_t C, int64_t G, int64_t ld, int64_t cell0, int64_t C_out,
                    int64_t slab_genes, int dtype, vcy_stream stream);

/* Pooling straight from the loom's uint16 count layers (velocyto/constants.py:11): the size-normalised
 * inputs of knn_imputation are norm_factor[c] * counts[c,:] (analysis.py:546-549, 573-579), so
 *     out[c,:] = sum_p (w[p] * scale[indices[p]]) * counts[indices[p],:]
 * gives the same Sx / Ux while every gather moves 2-byte elements - or 1-byte ones: layers whose counts all fit a byte
 * may be held as uint8 (count_dtype VCY_U8; lossless, chosen per layer at upload).  countsS/countsU: (C, ld16) of
 * count_dtype, cells-major, ld16 % 16 == 0, zero padded; scaleS/scaleU: (C) fp64 per-cell factors (1 for size_norm=False);
 * out/out2: (C_out, ld_out) of `dtype`, ld_out % 16 == 0 (written with 16-byte stores, columns G..ld_out zeroed).
 * countsU/scaleU/out2 may all be NULL.  maximum as vcy_knn_pool. */
int vcy_knn_pool_counts(const void *countsS, const void *countsU, const double *scaleS, const double *scaleU, void *out,
                        void *out2, const int64_t *indptr, const int32_t *indices, const void *w, const int32_t *order,
                        int64_t C, int64_t G, int64_t ld16, int64_t ld_out, int64_t cell0, int64_t C_out, int maximum,
                        int64_t slab_genes, int count_dtype, int dtype, vcy_stream stream);

/* Atlas-scale pooling: the same product (neighbors.py:416-423 applied to S_sz = factor * S, analysis.py:546-549,
 * 1011-1019) gathered from a SPARSE count layer in CSR form - indptr (C + 1) int64, indices (nnz) int32 gene numbers
 * ascending inside a row, data (nnz) uint16 / uint8 counts (count_dtype) - for datasets whose layers do not fit dense
 * (BASELINE.json configs[4]; the reference loads every layer dense, analysis.py:59-61).  out (C_out, ld_out) dense rows of
 * `dtype`; g_indptr / g_indices / w: the kNN graph rows of the C_out output cells (entries name CSR rows);
 * scale (C) per-row size factors; maximum: np.maximum with the cell's own scaled counts (row cell0 + c).  Results are
 * bit-identical to vcy_knn_pool_counts on the densified layer.  slabptr: the table vcy_csr_slab_ptr fills,
 * C x (ceil(G / vcy_csr_slab_genes()) + 1) int32 = offsets inside each row of the first non-zero of every gene slab.
 * `indices` and `data` must hold at least 4 elements (the kernel reads quads of consecutive non-zeros with 16-byte loads,
 * pulled back to end inside the arrays): a layer with fewer than 4 non-zeros is padded by the caller; elements at or past
 * indptr[C] belong to no row and never enter a result.  */
int64_t vcy_csr_slab_genes(void);
int vcy_csr_slab_ptr(const int64_t *indptr, const int32_t *indices, int32_t *slabptr, int64_t C, int64_t G, vcy_stream stream);
int vcy_knn_pool_csr(const int64_t *indptr, const int32_t *indices, const void *data, const int32_t *slabptr, const double *scale,
                     void *out, const int64_t *g_indptr, const int32_t *g_indices, const void *w, const int32_t *order, int64_t C,
                     int64_t G, int64_t ld_out, int64_t cell0, int64_t C_out, int maximum, int count_dtype, int dtype, vcy_stream stream);

/* Exact Euclidean kNN in a low-dimensional space (what sklearn NearestNeighbors provides to
 * neighbors.knn_distance_matrix :363-376, BalancedKNN.fit/kneighbors :239-243,282 and
 * analysis.py:1547-1549).  xt: (P, ldx) TRANSPOSED coordinates (feature-major) of all C
 * points, fp32.  x64: (C, P) row-major fp64 coordinates for the exact re-rank.
 * For queries q0..q0+Q-1 writes the k nearest (the query itself excluded when include_self == 0,
 * an ordinary distance-0 candidate otherwise), nearest first, ties by index: idx (Q,k) int32,
 * dist (Q,k) fp64.
 * Any k < C: candidate lists up to ~4k entries are sorted in LDS, larger ones in the workspace.
 * workspace: vcy_knn_workspace_bytes(C, Q, k) bytes.                                     */
size_t vcy_knn_workspace_bytes(int64_t C, int64_t Q, int64_t k);
/* 1 when a search of k neighbours among C points takes the kernel that keeps its candidates in registers (k + 8 <= 128,
 * C <= 261 632): its workspace is one scratch row per 8 queries, so all queries fit one launch; 0: the (Q, C) distance
 * rows are materialised and callers walk the queries in blocks.                                                        */
int vcy_knn_row_free(int64_t C, int64_t k);
int vcy_knn_search(const float *xt, const double *x64, int32_t *idx, double *dist, void *workspace,
                   int64_t C, int64_t P, int64_t ldx, int64_t q0, int64_t Q, int64_t k, int include_self,
                   vcy_stream stream);

/* The same search for EXTERNAL query points (sklearn's NearestNeighbors.fit(points).kneighbors(X), as called by
 * calculate_grid_arrows analysis.py:1793-1795 and Diffusion.compute_transition_matrix2 diffusion.py:40-41):
 * qt (P, ldq) feature-major fp32 and q64 (>= q0+Q, P) row-major fp64 coordinates of the queries; for queries
 * q0..q0+Q-1 the k nearest of the C points, nearest first, ties by index.                                   */
int vcy_knn_query(const float *xt, const double *x64, const float *qt, const double *q64, int64_t ldq, int32_t *idx,
                  double *dist, void *workspace, int64_t C, int64_t P, int64_t ldx, int64_t q0, int64_t Q, int64_t k,
                  vcy_stream stream);

/* neighbors.balance_knn_loop / balance_knn_loop_constrained (neighbors.py:11-140): the
 * sequential greedy in-degree-capped selection.  HOST function on HOST pointers (the
 * reference runs it as numba-compiled scalar code; it is O(C * sight) integer work with a
 * loop-carried dependence).  groups_host may be NULL.                                     */
int vcy_balance_knn_host(const int64_t *dsi_host, const double *dist_host, const int64_t *lsi_host,
                         const int64_t *groups_host, int64_t n, int64_t K, int64_t maxl, int64_t k,
                         int return_distance, double *dist_new_host, int64_t *dsi_new_host, int64_t *l_host);

/* The same selection on int32 sight lists (the type vcy_knn_search returns) without distances on the host: pos_new_host (n, k+1)
 * receives the position of every selected neighbour in its cell's sight list (-1: column 0 and padded slots), so that the caller
 * gathers the distances where they live.  For the reference's default sight = whole dataset (analysis.py:985-988) this keeps the
 * host side at n * K * 4 bytes instead of n * K * 16.                                                                          */
int vcy_balance_knn_host32(const int32_t *dsi_host, const int64_t *lsi_host, const int64_t *groups_host, int64_t n, int64_t K,
                           int64_t maxl, int64_t k, int32_t *pos_new_host, int64_t *dsi_new_host, int64_t *l_host);

/* ---------------------------------------------------------------- stage B: gamma fits
 * estimation.fit_slope + _fit1_slope (estimation.py:173-188, 267-279): per gene
 *     gamma = max(0, sum_c x*y / sum_c x*x), NaN if x == 0 everywhere, 0 if y == 0 everywhere;
 * Y = unspliced, X = spliced, (C, ld) cells-major; gamma: (G) float32 like the reference.
 * workspace: vcy_fit_workspace_bytes(G) bytes.                                           */
size_t vcy_fit_workspace_bytes(int64_t G);
int vcy_fit_slope(const void *Y, const void *X, float *gamma, void *workspace, int64_t C, int64_t G,
                  int64_t ld, int dtype, vcy_stream stream);

/* The two halves of vcy_fit_slope for cell-sharded runs: each rank reduces its own cells to
 * moments (3, G) fp64 = [sum x*x, sum x*y, sum y*y], the ranks all-reduce(sum) that 3*G vector
 * (RCCL), and every rank finishes the fit from the global moments.                        */
int vcy_fit_slope_moments(const void *Y, const void *X, double *moments, void *workspace, int64_t C, int64_t G,
                          int64_t ld, int dtype, vcy_stream stream);
int vcy_fit_slope_from_moments(const double *moments, float *gamma, int64_t G, vcy_stream stream);

/* Per-gene raw moments over cells, (5, G) fp64 = [sum x, sum y, sum x*x, sum x*y, sum y*y] (x = X, y = Y): the paired row
 * correlation of filter_genes_by_phase_portrait (analysis.py:1285-1288, 1307).  workspace: vcy_fit_workspace_bytes(G).   */
int vcy_gene_moments(const void *Y, const void *X, double *moments, void *workspace, int64_t C, int64_t G, int64_t ld,
                     int dtype, vcy_stream stream);

/* Per-gene statistics over cells for the filters upstream of the path, one streaming pass:
 * stats (4, G) fp64 = [sum x, sum x*x, count(x > 0), max x] of x = clip(M[c,g] * cell_scale[c], lo[g], hi[g]) over the cells with
 * cell_mask[c] != 0.  cell_scale (C, fp64), lo/hi (G, fp64; together) and cell_mask (C, uint8) may be NULL.
 * Replaces the numpy reductions of score_detection_levels (analysis.py:466-474: S.sum(1), (S > 0).sum(1)), score_cv_vs_mean
 * (analysis.py:260-272: detection, mean, std(ddof=1), np.clip winsorising) and clusters_stats (estimation.py:380-387).
 * dtype: VCY_F32 / VCY_F64 / VCY_U16 / VCY_U8 (raw loom counts, ld in elements).  workspace: vcy_gene_stats_workspace_bytes(G).  */
int64_t vcy_gene_stats_workspace_bytes(int64_t G);
int vcy_gene_stats(const void *M, const double *cell_scale, const double *lo, const double *hi, const uint8_t *cell_mask,
                   double *stats, void *workspace, int64_t C, int64_t G, int64_t ld, int dtype, vcy_stream stream);

/* Whole-matrix scale facts of a cells-major matrix, one streaming pass: out3 (fp64, device) = [sum |x|, smallest non-zero |x|
 * (+inf when every entry is zero; a denormal counts as non-zero), number of non-zero entries] over the G logical columns.
 * What the host layer needs to pick `rules` for the partial sqrt kernels (VCY_RULES_PARTIAL_NOPSC drops the pseudocount of
 * speedboosted.pyx:372-378, which is only admissible on a matrix of ordinary scale without sub-normal-range entries).
 * dtype VCY_F32 / VCY_F64.  workspace: vcy_abs_stats_workspace_bytes().                                                      */
int64_t vcy_abs_stats_workspace_bytes(void);
int vcy_abs_stats(const void *M, double *out3, void *workspace, int64_t C, int64_t G, int64_t ld, int dtype, vcy_stream stream);

/* Per-gene order statistics over cells with numpy.percentile's linear interpolation
 * (analysis.py:1183-1218 use np.percentile(M, q, axis=1)).  M: (C, ld) cells-major.
 * qs_host: nq percentiles in [0,100] (host array).  out: (nq, G) fp64.
 * If scale_a/scale_b (G, fp64) are non-NULL the statistic is taken of
 *     M[c,g]/scale_a[g] + M2[c,g]/scale_b[g]      (the maxmin_diag sum, analysis.py:1203-1206).
 * Conditional percentiles (estimation.py:200-202, 222, 229-231, 255: y[x > percentile(x,90)],
 * y[x <= percentile(x,1)]): mask_mode 1 keeps cells with mask_src[c,g] > mask_thr[g], 2 keeps
 * mask_src[c,g] <= mask_thr[g], 0 = no mask (mask_src, mask_thr NULL); a gene with no cell kept
 * gives NaN.
 * workspace: vcy_quantile_workspace_bytes(C, G).                                         */
size_t vcy_quantile_workspace_bytes(int64_t C, int64_t G);
int vcy_gene_quantiles(const void *M, const void *M2, const double *scale_a, const double *scale_b,
                       const void *mask_src, const double *mask_thr, int mask_mode,
                       const double *qs_host, int nq, double *out, void *workspace, int64_t C, int64_t G,
                       int64_t ld, int dtype, vcy_stream stream);

/* estimation.fit_slope_weighted_offset / fit_slope_weighted / fit_slope_offset with the
 * binary or dense weights of VelocytoLoom.fit_gammas (estimation.py:191-264, 300-366;
 * analysis.py:1179-1257), solved exactly: weighted moments per gene in one pass, then the
 * closed-form box-constrained least squares the reference hands to L-BFGS-B / Brent.
 *   weight_mode 0: W given densely (C, ld) of `dtype`
 *   weight_mode 1: W = (Z <= down[g]) | (Z >= up[g]) with Z = M/scale_a + M2/scale_b
 *                  (scale pointers NULL -> Z = M), thresholds (G) fp64  -- "maxmin*" weights
 *   weight_mode 2: unweighted (W = 1)
 *   fit_offset != 0 fits (gamma, q) in the box [lo_gamma, up_gamma[g]] x [0, 2*sum(yw)/sum(w)];
 *   fit_offset == 0 fits gamma only in [lo_gamma, up_gamma[g]] with fixed offset q_fixed[g]
 *   (NULL -> 0).  up_gamma may be NULL (-> up_gamma_default).
 * Outputs gamma, q, R2: (G) float32 (R2 unweighted, -1e16 when non-finite; :354-363).     */
int vcy_fit_weighted(const void *Y, const void *X, int weight_mode, const void *W, const void *M, const void *M2,
                     const double *scale_a, const double *scale_b, const double *down, const double *up,
                     int fit_offset, int box_q, double lo_gamma, double up_gamma_default, const double *up_gamma,
                     const double *q_fixed, float *gamma, float *q, float *R2, void *workspace, int64_t C,
                     int64_t G, int64_t ld, int dtype, vcy_stream stream);

/* ---------------------------------------------------------------- stage C: velocity chain
 * predict_U -> calculate_velocity -> calculate_shift -> extrapolate_cell_at_t and the
 * `dmat` transform of estimate_transition_prob, fused (analysis.py:1321-1439, 1538,
 * 1575-1601).  Any output pointer may be NULL (not materialised).
 *   Upred   = gamma*Sx_sz + q
 *   velocity= Ux_sz - Upred            (|v| < eps_thr[g] -> 0 when eps_thr != NULL)
 *   delta_S = dt_shift*velocity                       (assumption 0, constant_velocity)
 *           = Sx*e^{-g dt} + (1-e^{-g dt})*max(Ux-q,0)/g - Sx   (assumption 1, constant_unspliced)
 *   Sx_sz_t = Sx_sz + dt_extrap*delta_S, clipped at 0 when clip != 0
 *   dmat    = sign(D)*sqrt(|D|+psc) | sign(D)*log10(|D|+psc) | D,  D = (Sx_sz + used_dt*delta_S) - Sx_sz
 * gamma, q: (G) float32 as the reference stores them (q may be NULL).                      */
int vcy_velocity_chain(const void *Sx_sz, const void *Ux_sz, const float *gamma, const float *q,
                       const double *eps_thr, void *Upred, void *velocity, void *delta_S, void *Sx_sz_t,
                       void *dmat, int64_t C, int64_t G, int64_t ld, double dt_shift, double dt_extrap,
                       double used_dt, int assumption, int clip, int transform, double psc, int dtype,
                       vcy_stream stream);

/* One stage of the same chain from its STORED predecessor - calculate_velocity reads self.Upred (analysis.py:1369),
 * calculate_shift self.velocity (:1399), extrapolate_cell_at_t self.delta_S (:1429-1431) - so that a matrix the user has
 * edited between two calls propagates as it does in the reference:  out = a * x + b * y  (y may be NULL: out = a * x),
 * then |out| < zero_below[g] -> 0 (zero_below (G) doubles or NULL: the eps rule, :1377-1379) and max(out, 0) when clip.
 * Products and sum are rounded separately (numpy's evaluation order).  Cells-major (C, ld) matrices of `dtype`.          */
int vcy_lincomb(const void *x, const void *y, void *out, double a, double b, const double *zero_below, int clip, int64_t C,
                int64_t G, int64_t ld, int dtype, vcy_stream stream);

/* The non-default weight constructions of VelocytoLoom.fit_gammas (analysis.py:1182-1192,
 * 1208-1219), materialised densely for vcy_fit_weighted(weight_mode 0).  Per-gene fp64 vectors:
 *   mode 0 "sum"   W = S/pa + U/pb            mode 1 "prod"  W = (S/pa)*(U/pb)
 *   mode 2 "maxmin_weighted"  R = (clip(S,pa,pb)-pa)/(pb-pa), W = 0.5 (R^power + (1-R)^power)
 *   mode 3 "maxmin_double"    W = [Z<=pa | Z>=pb] + [S<=pc | S>=pd],  Z = S/sa + U/sb          */
int vcy_gamma_weights(const void *S, const void *U, void *W, const double *pa, const double *pb, const double *pc,
                      const double *pd, const double *sa, const double *sb, int64_t C, int64_t G, int64_t ld, int mode,
                      double power, int dtype, vcy_stream stream);

/* ---------------------------------------------------------------- pre-step a1: normalisation
 * VelocytoLoom._normalize_S/_normalize_U/_normalize_Sx/_normalize_Ux (analysis.py:535-631):
 * cell_size = M.sum(over genes) per cell;  out_sz = factor[c]*M (non-finite -> 0 when
 * fix_nonfinite, :580);  out_norm = log2(out_sz + pcount).  out_sz / out_norm may be NULL;
 * factor NULL = 1 (size=False).                                                            */
int vcy_row_sums(const void *M, double *out, int64_t C, int64_t G, int64_t ld, int dtype, vcy_stream stream);
int vcy_scale_log(const void *M, const double *factor, void *out_sz, void *out_norm, int64_t C, int64_t G, int64_t ld,
                  double pcount, int fix_nonfinite, int dtype, vcy_stream stream);

/* ---------------------------------------------------------------- estimate_transition_prob helpers
 * dmat (and, for logratio, the transformed expression e_out) from a STORED delta_S
 * (analysis.py:1538, 1575-1601, 1637-1663): hi_dim_t = hi_dim + used_dt*delta_S;
 * mode 0 linear: dmat = hi_dim_t - hi_dim; 1 sqrt / 2 log10: sign(D) f(|D| + psc);
 * 3 logratio: e_out = log2(hi_dim + psc), dmat = log2(|hi_dim_t| + psc) - e_out.          */
int vcy_delta_transform(const void *hi_dim, const void *delta_S, void *dmat, void *e_out, int64_t C, int64_t G, int64_t ld,
                        double used_dt, int mode, double psc, int dtype, vcy_stream stream);
/* permute_rows_nsign (analysis.py:2407-2420, called at :1540-1541 for the randomised control): out[c, g] = +- in[pi_g(c), g] with
 * an independent pseudo-random permutation pi_g of the cells and independent random signs per gene, both functions of (seed, g).
 * The reference draws them from numba's RNG stream: statistical parity, not the same numbers.  in != out; padding columns of out
 * are zeroed.  workspace_a / workspace_b (optional, both or neither; vcy_permute_rows_nsign_workspace_bytes each - any two
 * buffers of a matrix's size will do, e.g. the ones the caller is about to fill with the two dmat transforms): the shuffle runs on
 * a gene-major copy, where a gene's values are contiguous - three passes over the matrix instead of a 64-byte sector per value;
 * NULL: one gather on the cell-major matrix.  Same result either way.                           */
size_t vcy_permute_rows_nsign_workspace_bytes(int64_t C, int64_t G, int dtype);
int vcy_permute_rows_nsign(const void *in, void *out, void *workspace_a, void *workspace_b, int64_t C, int64_t G, int64_t ld,
                           uint64_t seed, int dtype, vcy_stream stream);
/* np.fill_diagonal(corrcoef, 0) and corrcoef[isnan] = nan_to (analysis.py:1604-1612, 1666-1668) on
 * the compact (C_out, nrndm) form; nan_count (device int, may be NULL) counts the NaNs seen.  */
int vcy_corr_fixup(void *vals, const int32_t *ixs, int64_t cell0, int64_t C_out, int64_t nrndm, int zero_self, int fix_nan,
                   double nan_to, int *nan_count, int dtype, vcy_stream stream);

/* ---------------------------------------------------------------- stage E: calculate_embedding_shift
 * (analysis.py:1670-1733) in neighbour-list form.  corr, ixs: (C_out, n) compact correlations and
 * the embedding_knn non-zeros of each row; embedding: (C, edim <= 4) fp64.
 *   tp[c,k]    = exp(corr/sigma) / sum_k exp(corr/sigma)          (transition_prob non-zeros)
 *   wdiff[c,k] = tp[c,k] - 1/n                                    (pooling weights of :1716)
 *   delta_embedding[c,:] = sum_k wdiff[c,k] * unit(emb[ixs[c,k]] - emb[c])     (:1704-1712)
 * tp / wdiff may be NULL.  expression_scaling = vcy_knn_pool with wdiff, then
 * vcy_row_cosproj(delta_S, estim_delta): out[c] = <a_c, b_c> / |b_c|  (:1717).              */
int vcy_transition_prob(const void *corr, const int32_t *ixs, const double *embedding, int edim, void *tp, void *wdiff,
                        double *delta_embedding, int64_t cell0, int64_t C_out, int64_t n, double sigma_corr, int dtype,
                        vcy_stream stream);
int vcy_row_cosproj(const void *A, const void *B, double *out, int64_t C, int64_t G, int64_t ld, int dtype, vcy_stream stream);
/* The expression scaling of calculate_embedding_shift in ONE launch and without the (genes, cells) estimates
 * (analysis.py:1714-1719 and, with the *_rndm arguments, :1726-1731 for the randomised control):
 *   estim[c, :]   = sum_k wdiff[c, k] * hi_dim[ixs[c, k], :]                     (hi_dim @ transition_prob.T - hi_dim @ (knn / n).T)
 *   cos_proj[c]   = <delta_S[c, :], estim[c, :]> / |estim[c, :]|                 (0 / 0 = NaN as in the reference)
 * hi_dim, delta_S[, delta_S_rndm]: (C, ld) cells-major of `dtype`; ixs, wdiff[, wdiff_rndm]: (C_out, n) - the outputs of
 * vcy_transition_prob for the real [and the control] correlations over the same neighbour lists; order: schedule of the C_out
 * cells (NULL = natural; results do not depend on it, cells adjacent in it share the gathers of their common neighbours);
 * cos_proj[, cos_proj_rndm]: (C_out) fp64.  n <= vcy_embedding_scaling_max_neighbors() (256); wider lists return
 * VCY_ERR_UNSUPPORTED and are handled by vcy_knn_pool_w2 + vcy_row_cosproj.  A member's neighbours are added in ascending
 * cell number, the sums over genes in a fixed order: reproducible run to run.                                                    */
int vcy_embedding_scaling_max_neighbors(void);
int vcy_embedding_scaling(const void *hi_dim, const void *delta_S, const void *delta_S_rndm, const int32_t *ixs, const void *wdiff,
                          const void *wdiff_rndm, const int32_t *order, double *cos_proj, double *cos_proj_rndm, int64_t C, int64_t G,
                          int64_t ld, int64_t C_out, int64_t n, int dtype, vcy_stream stream);

/* ---------------------------------------------------------------- stage F: prepare_markov
 * VelocytoLoom.prepare_markov (analysis.py:1818-1863) with cells_ixs=None: dense (n, n) Markov
 * matrix tr = rownorm(0.8 rownorm(P*K_D, diag := row max) + 0.2 rownorm(K_W)).  P as CSR
 * (indptr, indices, fp64 values): transition_prob for "forward", its transpose for "backwards".
 * embedding (n, edim <= 4) fp64; tr (n, n) of `dtype`.                                      */
int vcy_prepare_markov(const int64_t *indptr, const int32_t *indices, const double *pval, const double *embedding, int edim,
                       void *tr, int64_t n, double sigma_D, double sigma_W, int dtype, vcy_stream stream);

/* The same chain in factored form, for when the dense (n, n) matrix is the problem (10-20 GB at 50 000 cells, read once per step):
 * tr[c,j] = (0.2 K_W(c,j) / kw[c] + s[c,j]) / tot[c] with s sparse (pattern of P plus the diagonal) and K_W the Gaussian of the
 * embedding distance (analysis.py:1853-1862 regrouped).  vcy_prepare_markov_factored writes sval (nnz of P, CSR order; 0 where P
 * stores a diagonal entry), sdiag (n), kw (n), tot (n) and es (n, edim) of `compute_dtype` (VCY_F32 / VCY_F64) = embedding scaled so
 * that K_W = g exp2(-|es_c - es_j|^2).  vcy_diffuse_step_factored is one step y = x . tr (+ accum += y) of Diffusion.diffuse
 * (diffusion.py:93-105) from those factors: colptr / rowidx / scsc = s including its diagonal in CSC form; the Gauss transform is
 * evaluated on the fly in `compute_dtype` (8 terms folded before they reach the fp64 accumulator), the rest in fp64.
 * workspace: vcy_markov_factored_workspace_bytes(n).  prepared: 0 for a step that starts from an arbitrary x; 1 when x is the y of
 * the previous step ON THE SAME WORKSPACE (a loop of steps): that step's fold has already written x / tot and its scaled copy there,
 * and the scaling launch is skipped (loops of thousands of small steps are launch-bound).                                        */
size_t vcy_markov_factored_workspace_bytes(int64_t n);
int vcy_prepare_markov_factored(const int64_t *indptr, const int32_t *indices, const double *pval, const double *embedding, int edim,
                                double *sval, double *sdiag, double *kw, double *tot, void *es, int64_t n, double sigma_D, double sigma_W,
                                int compute_dtype, vcy_stream stream);
int vcy_diffuse_step_factored(const double *x, double *y, double *accum, const int64_t *colptr, const int32_t *rowidx, const double *scsc,
                              const double *tot, const double *kw, const void *es, int edim, double sigma_W, void *workspace, int64_t n,
                              int prepared, int compute_dtype, vcy_stream stream);

/* The factored step when K_W is narrow against the extent of the embedding (prepare_markov is usually given a sigma_W of a grid step,
 * analysis.py:1818-1863): terms below 2^-cut of their weight are left out of the Gauss transform.  The caller sorts the cells along a
 * space-filling curve of the embedding: es_sorted (n, edim) = the scaled coordinates of vcy_prepare_markov_factored in that order,
 * rank (n) = position of every cell in it, order (n) = the cell at every position.  vcy_markov_cull_boxes fills `boxes` (vcy_markov_cull_boxes_bytes) with the bounding boxes
 * of runs of 32 consecutive sorted cells; vcy_diffuse_step_factored_culled skips the runs whose box is farther than sqrt(cut) (in the
 * scaled units, where K_W = g exp2(-d^2)) from the box of a workgroup's targets.  Same arguments and result as
 * vcy_diffuse_step_factored otherwise; per target at most n max(x / tot) 2^-cut is dropped (cut 48 / 72 for f32 / f64 compute). */
size_t vcy_markov_cull_boxes_bytes(int64_t n, int edim, int compute_dtype);
int vcy_markov_cull_boxes(const void *es_sorted, void *boxes, int64_t n, int edim, int compute_dtype, vcy_stream stream);
int vcy_diffuse_step_factored_culled(const double *x, double *y, double *accum, const int64_t *colptr, const int32_t *rowidx,
                                     const double *scsc, const double *tot, const double *kw, const void *es_sorted, const int32_t *rank,
                                     const int32_t *order, const void *boxes, int edim, double sigma_W, double cut, void *workspace,
                                     int64_t n, int prepared, int compute_dtype, vcy_stream stream);

/* ---------------------------------------------------------------- stage F: Diffusion.diffuse step
 * (diffusion.py:93-105): y = x . tr, optionally accum += y (path_integral).  tr dense row-major
 * (n, n) of `dtype`, or CSC (column pointers / row indices / values) for sparse matrices;
 * x, y, accum fp64 (n).  workspace: vcy_diffuse_workspace_bytes(n).                        */
size_t vcy_diffuse_workspace_bytes(int64_t n);
int vcy_diffuse_step_dense(const void *tr, const double *x, double *y, double *accum, void *workspace, int64_t n, int dtype,
                           vcy_stream stream);
int vcy_diffuse_step_csc(const int64_t *colptr, const int32_t *rowidx, const void *val, const double *x, double *y,
                         double *accum, int64_t n, int dtype, vcy_stream stream);

/* ---------------------------------------------------------------- upstream caller: perform_PCA's dense contraction (f64 MFMA)
 * VelocytoLoom.perform_PCA (analysis.py:678-702) fits sklearn.decomposition.PCA on S_norm[pca_genes].T: centre every gene, then
 * the spectral decomposition of the centred (cells x genes) matrix.  The device path takes the covariance route; its one dense
 * product is
 *     vcy_gram:     gram[i][j] = sum_c (X[c][i] - mean[i]) (X[c][j] - mean[j])       G x G fp64, row pitch ldg, both triangles
 *     vcy_gram_tn:  out[i][j]  = sum_c (X[c][i] - mean[i]) Y[c][j]                   G x L fp64 (the block product X_c^T (X_c Z)
 *                                                                                    of the subspace iteration; Y (C, ldy) fp64)
 * X: (C, ld) cells-major of `dtype` (16-byte aligned rows); mean: (G) fp64 device or NULL (no centring); the contraction runs on
 * v_mfma_f64_16x16x4_f64 with fp64 accumulation in a fixed order (results are reproducible run to run).  workspace:
 * vcy_gram_workspace_bytes(C, G, L, symmetric) bytes (symmetric = 1 for vcy_gram, where L is ignored) - partial tiles when the
 * cells are split over several workgroups per output tile; vcy_col_means uses the same workspace.
 * vcy_col_means: mean[g] = sum_c X[c][g] / C in fp64, fixed summation order (sklearn's X.mean(axis=0)).                          */
size_t vcy_gram_workspace_bytes(int64_t C, int64_t G, int64_t L, int symmetric);
int vcy_col_means(const void *X, double *mean, void *workspace, int64_t C, int64_t G, int64_t ld, int dtype, vcy_stream stream);
int vcy_gram(const void *X, const double *mean, double *gram, void *workspace, int64_t C, int64_t G, int64_t ld, int64_t ldg, int dtype,
             vcy_stream stream);
int vcy_gram_tn(const void *X, const double *mean, const double *Y, double *out, void *workspace, int64_t C, int64_t G, int64_t L, int64_t ld,
                int64_t ldy, int64_t ldo, int dtype, vcy_stream stream);

/* The products of the same caller that contract over the GENES (analysis.py:678-702 through sklearn's PCA: the projection of the centred matrix on
 * a thin block - every pass of the subspace iteration, and `transform`, the scores pcs = (X - mean) components^T - and, with fewer cells than
 * genes, the cells' own Gram matrix):
 *     out[i][j] = sum_{g < K} A[i][g] B[j][g]  -  row_corr[i]  -  col_corr[j]  +  c0          out (M, ldo) fp64
 * A: (M, lda) of dtype_a (VCY_F32 / VCY_F64), B: (N, ldb) of dtype_b (dtype_a, or VCY_F64 beside an f32 A), both walked along their contiguous
 * dimension; row_corr (M) / col_corr (N) fp64 device vectors or NULL.  Centring is algebra on the corrections, not a pass over the matrix:
 * (X - m) Z = X Z - 1 (m Z); (X - m)(X - m)^T = X X^T - a 1^T - 1 a^T + m.m with a = X m - the rows are read as stored, once.
 * v_mfma_f64_16x16x4_f64, 128 x 64 tiles, slabs DMA'd into LDS (global_load_lds_dwordx4), fp64 accumulation in a fixed order.  Both row pitches must
 * hold whole slabs of 16 (f64 A) / 32 (f32 A) genes with ZEROS beyond K (the cells-major layout's padding), rows 16-byte aligned.            */
int vcy_gemm_nt(const void *A, const void *B, const double *row_corr, const double *col_corr, double c0, double *out, int64_t M, int64_t N, int64_t K,
                int64_t lda, int64_t ldb, int64_t ldo, int dtype_a, int dtype_b, vcy_stream stream);

/* ---------------------------------------------------------------- upstream callers: epsilon-SVR, RBF kernel, scalar inputs
 * The noise models of score_cv_vs_mean (analysis.py:280-282, 324-326: sklearn.svm.SVR(gamma=150/G).fit(log2 mean, log2 CV), one point
 * per gene) and adjust_totS_totU (analysis.py:844-851: SVR(C=100, kernel="rbf", gamma=1e-6) on per-cell totals).  scikit-learn
 * delegates to libsvm: SMO with second-order working-set selection, stopping at a KKT violation < tol; vcy_svr_rbf_fit runs the
 * same iteration on the device and agrees with it to the solver tolerance (not bit for bit: libsvm rounds kernel rows to float
 * and shrinks).  x, t: (n) fp64 device.  coef (n) = alpha - alpha* (0 for non-support points), intercept (1),
 * info (4, int32) = [SMO steps, converged, grid barrier failed, workgroups used].  max_iter <= 0: libsvm's own cap.
 * workspace: vcy_svr_workspace_bytes(n).
 * vcy_svr_rbf_predict: out[q] = sum_k coef[k] exp(-gamma (xq[q] - x[k])^2) + intercept[0]   (SVR.predict), xq / out: (m).     */
int64_t vcy_svr_workspace_bytes(int64_t n);
int vcy_svr_rbf_fit(const double *x, const double *t, double *coef, double *intercept, int32_t *info, void *workspace, int64_t n,
                    double C, double epsilon, double gamma, double tol, int64_t max_iter, vcy_stream stream);
int vcy_svr_rbf_predict(const double *x, const double *coef, const double *intercept, const double *xq, double *out, int64_t n,
                        int64_t m, double gamma, vcy_stream stream);

/* ---------------------------------------------------------------- host helper: neighbour sampling of estimate_transition_prob
 * analysis.py:1561-1564 draws, per cell, np.random.choice(n, size, replace=False, p=p) from numpy's legacy global RNG.  This is
 * RandomState.choice(replace=False, p) restated over a pool of uniforms the caller drew from the same RandomState in one call
 * (host pointers, no device work): out (cells, size) int64 gets what the per-cell calls would have returned, *cells_done the
 * number of cells whose draws fitted in the pool and *consumed the uniforms they used (advance the RandomState by that).     */
int vcy_choice_stream_host(const double *pool, int64_t pool_len, const double *p, int64_t n, int64_t size, int64_t cells,
                           int64_t *out, int64_t *cells_done, int64_t *consumed);

#ifdef __cplusplus
}
#endif
#endif /* VELOCYTO_HIP_H */
