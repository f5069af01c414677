#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE ITSELF (build container only).

The reference (velocyto.py @ /root/reference) ships no tests and no golden vectors
(SURVEY.md section 4), so the pins for this path are outputs of the reference's own code
executed here on small seeded inputs:

  * ``velocyto/speedboosted.pyx`` compiled with the reference's flags by
    ``oracle/build_ref.py`` (-> oracle/_ref/, never committed);
  * ``velocyto/estimation.py``, ``neighbors.py``, ``diffusion.py``, ``analysis.py`` imported
    *from /root/reference* (nothing is copied) under a synthetic package object so that the
    package ``__init__`` (which needs pysam/loompy/numba, absent here) does not run.

Shims applied in THIS process only (reference files untouched), all forced by library
version drift and listed in SURVEY.md section 8(c):
  numba      -> identity ``jit`` stub (the jitted loops run as plain Python);
  loompy/h5py-> empty stub modules (only the ctor / serialization touch them);
  np.NAN     -> np.nan (removed in numpy 2);
  np.stack   -> accepts a generator (analysis.py:1561);
  scipy.sparse matrices -> ``.A`` property (removed in scipy 1.14);
  scipy.sparse.csr -> old module alias used in type hints (neighbors.py:379,385).
``VelocytoLoom`` objects are created with ``__new__`` and fed arrays directly (the loom
constructor needs loompy); Sx/Ux-derived arrays are made C-contiguous before
``estimate_transition_prob`` (the reference's own F-order trap).

Only numeric inputs/outputs are written; the fixtures are data, not code.
Usage:  python tests/golden/make_golden.py            (re-creates every fixture)
"""
import importlib
import importlib.machinery
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/velocyto"


def load_reference():
    import scipy.sparse as sp
    import matplotlib
    matplotlib.use("Agg")
    if not hasattr(np, "NAN"):
        np.NAN = np.nan
    _stack = np.stack

    def stack(arrays, *a, **k):
        if isinstance(arrays, types.GeneratorType):
            arrays = list(arrays)
        return _stack(arrays, *a, **k)
    np.stack = stack
    for cls in (sp.csr_matrix, sp.csc_matrix, sp.coo_matrix, sp.lil_matrix):
        if not hasattr(cls, "A"):
            cls.A = property(lambda self: self.toarray())
    if not hasattr(sp, "csr") or not hasattr(getattr(sp, "csr", None), "csr_matrix"):
        sp.csr = types.SimpleNamespace(csr_matrix=sp.csr_matrix)

    numba = types.ModuleType("numba")

    def jit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f
    numba.jit = jit
    numba.njit = jit
    sys.modules["numba"] = numba
    for name in ("loompy", "h5py"):
        sys.modules.setdefault(name, types.ModuleType(name))

    pkg = types.ModuleType("velocyto")
    pkg.__path__ = [REF]
    sys.modules["velocyto"] = pkg
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from build_ref import build
    so = build()
    loader = importlib.machinery.ExtensionFileLoader("velocyto.speedboosted", so)
    spec = importlib.util.spec_from_loader("velocyto.speedboosted", loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    sys.modules["velocyto.speedboosted"] = mod
    est = importlib.import_module("velocyto.estimation")
    nb = importlib.import_module("velocyto.neighbors")
    dif = importlib.import_module("velocyto.diffusion")
    ana = importlib.import_module("velocyto.analysis")
    return est, nb, dif, ana


def synth_counts(rng, G, C, n_clusters=4):
    """Small synthetic spliced/unspliced counts with some structure (uint16 like a loom)."""
    t = rng.random(C)
    alpha = rng.lognormal(0, 1, G)
    gamma = rng.lognormal(-0.5, 0.5, G)
    on = rng.random(G)
    u = alpha[:, None] * (t[None, :] > on[:, None] * 0.6) * (1 - np.exp(-3 * np.maximum(t[None, :] - on[:, None] * 0.6, 0)))
    s = u / gamma[:, None] * (1 - np.exp(-2 * np.maximum(t[None, :] - on[:, None] * 0.6, 0))) + 0.2 * alpha[:, None]
    size = rng.lognormal(0, 0.3, C)
    S = rng.poisson(4 * size[None, :] * s).astype(np.uint16)
    U = rng.poisson(2 * size[None, :] * (u + 0.05)).astype(np.uint16)
    return S, U, t


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def golden_coldeltacor(est):
    rng = np.random.default_rng(20180808)
    G, C, nr = 60, 40, 12
    e = rng.gamma(2.0, 1.0, (G, C))
    e[rng.random((G, C)) < 0.3] = 0.0       # exact zero differences between many cells
    e[:, 7] = e[:, 3]                       # identical cells -> zero-variance column (NaN / zero-rule)
    d = rng.normal(0, 1, (G, C))
    ixs = np.stack([rng.choice(C, nr, replace=False) for _ in range(C)])
    ixs[3, 0] = 7                           # make the identical pair appear
    ixs[5, 1] = 5                           # a cell listing itself
    out = dict(e=e, d=d, ixs=ixs)
    with np.errstate(all="ignore"):
        out["full_linear"] = est.colDeltaCor(e, d, threads=1)
        out["partial_linear"] = est.colDeltaCorpartial(e, d, ixs, threads=1)
        for psc in (1e-10, 1.0):
            tag = "a" if psc == 1e-10 else "b"
            out[f"full_sqrt_{tag}"] = est.colDeltaCorSqrt(e, d, threads=1, psc=psc)
            out[f"partial_sqrt_{tag}"] = est.colDeltaCorSqrtpartial(e, d, ixs, threads=1, psc=psc)
            out[f"full_log10_{tag}"] = est.colDeltaCorLog10(e, d, threads=1, psc=psc)
            out[f"partial_log10_{tag}"] = est.colDeltaCorLog10partial(e, d, ixs, threads=1, psc=psc)
    out["psc_a"] = np.float64(1e-10)
    out["psc_b"] = np.float64(1.0)
    save("coldeltacor", **out)


def golden_fits(est):
    rng = np.random.default_rng(20180809)
    G, C = 24, 90
    X = rng.gamma(1.5, 1.0, (G, C))
    gam = rng.lognormal(-0.5, 0.6, G)
    Y = np.maximum(gam[:, None] * X + 0.15 + rng.normal(0, 0.25, (G, C)), 0)
    X[0] = 0                                 # all-zero x  -> NaN
    Y[1] = 0                                 # all-zero y  -> 0
    Y[2] = 3.0 * X[2] + 2.0                  # y consistently above x (limit_gamma branch)
    Y[3] = np.maximum(0.05 * X[3] - 0.2, 0)  # pushes gamma/offset to the box edges
    W = (rng.random((G, C)) < 0.3).astype(float)
    W[4] = rng.random(C)                     # non-binary weights
    out = dict(Y=Y, X=X, W=W)
    out["fit_slope"] = est.fit_slope(Y, X)
    a, b = est.fit_slope_offset(Y, X)
    out["offset_m"], out["offset_q"] = a, b
    a, b = est.fit_slope_offset(Y, X, fixperc_q=True)
    out["offset_fix_m"], out["offset_fix_q"] = a, b
    for lg in (False, True):
        t = "lg" if lg else "nolg"
        a, r = est.fit_slope_weighted(Y, X, W, return_R2=True, limit_gamma=lg)
        out[f"weighted_{t}_m"], out[f"weighted_{t}_R2"] = a, r
        a, b, r = est.fit_slope_weighted_offset(Y, X, W, return_R2=True, limit_gamma=lg)
        out[f"woffset_{t}_m"], out[f"woffset_{t}_q"], out[f"woffset_{t}_R2"] = a, b, r
    a, b, r = est.fit_slope_weighted_offset(Y, X, W, fixperc_q=True, return_R2=True)
    out["woffset_fix_m"], out["woffset_fix_q"], out["woffset_fix_R2"] = a, b, r
    save("fits", **out)


def golden_neighbors(nb):
    rng = np.random.default_rng(20180810)
    C, P, G = 220, 8, 30
    centers = rng.normal(0, 4, (5, P))
    lab = rng.integers(0, 5, C)
    space = centers[lab] + rng.normal(0, 1, (C, P))
    space[11] = space[10]                    # duplicate cell -> zero distance neighbour
    data = rng.gamma(1.0, 2.0, (G, C))
    out = dict(space=space, data=data, groups=lab.astype(np.int64))
    k = 9
    knn = nb.knn_distance_matrix(space, k=k, mode="distance", n_jobs=1)
    out["knn_indices"] = knn.indices.reshape(C, k)
    out["knn_dist"] = knn.data.reshape(C, k)
    conn = (knn > 0).astype(float)
    conn.setdiag(3.0)
    w = nb.connectivity_to_weights(conn).tocsr()
    w.sort_indices()
    out["w_indptr"], out["w_indices"], out["w_data"] = w.indptr, w.indices, w.data
    out["w_diag"] = np.float64(3.0)
    out["convolved"] = np.ascontiguousarray(nb.convolve_by_sparse_weights(data, w))
    for tag, constraint in (("bal", None), ("balc", lab.astype(np.int64))):
        b = nb.BalancedKNN(k=k, sight_k=40, maxl=14, constraint=constraint, mode="distance", n_jobs=1)
        b.fit(space)
        g = b.kneighbors_graph(mode="distance")
        out[f"{tag}_dsi"], out[f"{tag}_dist"] = b.dsi, b.dist      # the sight graph fed to the greedy loop
        out[f"{tag}_dist_new"], out[f"{tag}_dsi_new"], out[f"{tag}_l"] = b.dist_new, b.dsi_new, b.l
        out[f"{tag}_graph_data"], out[f"{tag}_graph_indices"] = g.data, g.indices
    # sight exhaustion -> pad with self (neighbors.py:65-69)
    d2, i2, l2 = nb.knn_balance(out["bal_dsi"][:, :12], out["bal_dist"][:, :12], maxl=4, k=9)
    out["pad_dist_new"], out["pad_dsi_new"], out["pad_l"] = d2, i2, l2
    # (dist=None together with padding crashes in the reference itself - UnboundLocalError at
    #  neighbors.py:68 - so that combination has no defined behaviour and is not pinned.)
    d3, i3, l3 = nb.knn_balance(out["bal_dsi"], None, maxl=14, k=9)
    out["nd_dist_new"], out["nd_dsi_new"], out["nd_l"] = d3, i3, l3
    save("neighbors", **out)


def make_vlm(ana, S, U):
    vlm = ana.VelocytoLoom.__new__(ana.VelocytoLoom)
    vlm.S = S.astype(np.float64)
    vlm.U = U.astype(np.float64)
    vlm.A = np.zeros_like(vlm.S)
    vlm.ca = {"CellID": np.arange(S.shape[1])}
    vlm.ra = {"Gene": np.arange(S.shape[0])}
    vlm.initial_cell_size = vlm.S.sum(0)
    vlm.initial_Ucell_size = vlm.U.sum(0)
    return vlm


def golden_pipeline(ana, dif):
    rng = np.random.default_rng(20180811)
    G, C = 90, 160
    S, U, t = synth_counts(rng, G, C)
    keep = (S.sum(1) > 0) & (U.sum(1) > 0)
    S, U = S[keep], U[keep]
    out = dict(S=S, U=U)
    vlm = make_vlm(ana, S, U)
    vlm.normalize("both", size=True, log=True)
    out["S_sz"], out["U_sz"], out["S_norm"] = vlm.S_sz, vlm.U_sz, vlm.S_norm
    # pcs are an INPUT of the path (perform_PCA is out of scope): top singular vectors of centred S_norm
    Xc = vlm.S_norm.T - vlm.S_norm.T.mean(0)
    Uu, ss, _ = np.linalg.svd(Xc, full_matrices=False)
    vlm.pcs = Uu[:, :12] * ss[:12]
    vlm.ts = vlm.pcs[:, :2].copy()
    out["pcs"], out["ts"] = vlm.pcs, vlm.ts

    # balanced variant first (kept separately), then the default unbalanced graph used downstream
    vlm.knn_imputation(k=12, n_pca_dims=10, balanced=True, b_sight=48, b_maxl=20, n_jobs=1)
    out["bal_Sx"], out["bal_Ux"] = np.ascontiguousarray(vlm.Sx), np.ascontiguousarray(vlm.Ux)
    out["bal_knn_indices"], out["bal_knn_data"] = vlm.knn.indices, vlm.knn.data
    vlm.knn_imputation(k=12, n_pca_dims=10, diag=2.0, maximum=True, n_jobs=1)
    out["max_Sx"], out["max_Ux"] = np.ascontiguousarray(vlm.Sx), np.ascontiguousarray(vlm.Ux)
    vlm.knn_imputation(k=12, n_pca_dims=10, n_jobs=1)
    out["knn_indices"] = vlm.knn.indices.reshape(C, 12)
    out["knn_dist"] = vlm.knn.data.reshape(C, 12)
    out["Sx"], out["Ux"] = np.ascontiguousarray(vlm.Sx), np.ascontiguousarray(vlm.Ux)

    # plain fit (fit_slope) and the default weighted-offset fit
    vlm.fit_gammas(fit_offset=False, weighted=False)
    out["gammas_plain"] = vlm.gammas.copy()
    for wname in ("maxmin", "maxmin_double", "sum", "prod", "maxmin_weighted"):
        vlm.fit_gammas(weights=wname)
        out[f"gammas_{wname}"], out[f"q_{wname}"], out[f"R2_{wname}"] = vlm.gammas.copy(), vlm.q.copy(), vlm.R2.copy()
    vlm.fit_gammas(limit_gamma=True)
    out["gammas_lg"], out["q_lg"], out["R2_lg"] = vlm.gammas.copy(), vlm.q.copy(), vlm.R2.copy()
    vlm.fit_gammas(fit_offset=False, weighted=True)
    out["gammas_w"], out["R2_w"] = vlm.gammas.copy(), vlm.R2.copy()
    vlm.fit_gammas()
    out["gammas"], out["q"], out["R2"] = vlm.gammas.copy(), vlm.q.copy(), vlm.R2.copy()

    vlm.predict_U()
    vlm.calculate_velocity()
    vlm.calculate_shift(assumption="constant_unspliced", delta_t=0.7)
    out["delta_S_cu"] = np.ascontiguousarray(vlm.delta_S)
    vlm.calculate_shift(assumption="constant_velocity")
    vlm.extrapolate_cell_at_t(delta_t=1.0)
    out["Upred"], out["velocity"] = np.ascontiguousarray(vlm.Upred), np.ascontiguousarray(vlm.velocity)
    out["delta_S"], out["Sx_sz_t"] = np.ascontiguousarray(vlm.delta_S), np.ascontiguousarray(vlm.Sx_sz_t)

    for name in ("Sx", "Ux", "Sx_sz", "Ux_sz", "delta_S", "Sx_sz_t"):   # F-order trap
        setattr(vlm, name, np.ascontiguousarray(getattr(vlm, name)))

    with np.errstate(all="ignore"):
        for transform in ("sqrt", "log", "linear", "logratio"):
            vlm.estimate_transition_prob(hidim="Sx_sz", embed="ts", transform=transform, n_neighbors=40,
                                         knn_random=True, sampled_fraction=0.5, calculate_randomized=False, threads=1)
            out[f"corrcoef_{transform}"] = vlm.corrcoef.copy()
            if transform == "sqrt":
                out["sampling_ixs"] = vlm.sampling_ixs.copy()
                out["neigh_ixs"] = vlm.embedding_knn.indices.reshape(C, -1).copy()
        for transform in ("sqrt", "log", "linear"):
            vlm.estimate_transition_prob(hidim="Sx_sz", embed="ts", transform=transform, n_neighbors=40,
                                         knn_random=False, calculate_randomized=False, threads=1)
            out[f"corrcoef_full_{transform}"] = vlm.corrcoef.copy()
            if transform == "sqrt":
                out["full_knn_indices"] = vlm.embedding_knn.indices.reshape(C, -1).copy()
                vlm.calculate_embedding_shift(sigma_corr=0.05)
                out["full_transition_prob"] = vlm.transition_prob.copy()
                out["full_delta_embedding"] = vlm.delta_embedding.copy()
                out["full_scaling"] = vlm.scaling.copy()
        # the default path once more, then downstream stages E/F
        vlm.estimate_transition_prob(hidim="Sx_sz", embed="ts", transform="sqrt", n_neighbors=40,
                                     knn_random=True, sampled_fraction=0.5, calculate_randomized=False, threads=1)
        assert np.array_equal(vlm.corrcoef, out["corrcoef_sqrt"], equal_nan=True)
        vlm.calculate_embedding_shift(sigma_corr=0.05)
        out["transition_prob"] = vlm.transition_prob.copy()
        out["delta_embedding"] = vlm.delta_embedding.copy()
        out["scaling"] = vlm.scaling.copy()
        vlm.calculate_embedding_shift(sigma_corr=0.1, expression_scaling=False)
        out["delta_embedding_noscale"] = vlm.delta_embedding.copy()
        vlm.calculate_embedding_shift(sigma_corr=0.05)
        for direction in ("forward", "backwards"):
            vlm.prepare_markov(sigma_D=2.0, sigma_W=4.0, direction=direction)
            out[f"tr_{direction}"] = vlm.tr.toarray()
            vlm.run_markov(n_steps=50)
            out[f"diffused_{direction}"] = np.asarray(vlm.diffused).ravel()
    golden_next(vlm, dif, ana)
    p0 = rng.random(C)
    d = dif.Diffusion()
    out["diffuse_p0"] = p0
    out["diffuse_path_integral"] = np.asarray(d.diffuse(p0, vlm.tr, n_steps=7, mode="path_integral")).ravel()
    out["diffuse_time_evolution"] = np.asarray(d.diffuse(p0, vlm.tr, n_steps=7, mode="time_evolution")).ravel()
    save("pipeline", **out)


def golden_next(vlm, dif, ana):
    """SURVEY.md section 8(f) "next" rows, recorded from the same pipeline state: calculate_grid_arrows, the two
    Diffusion transition-matrix builders, filter_genes_by_phase_portrait."""
    from copy import deepcopy
    out = {}
    # Markov chain on a SUBSET of the cells (tutorial usage: prepare_markov(..., cells_ixs=...), analysis.py:1841-1863)
    sub = np.arange(3, vlm.S.shape[1], 2)
    out["markov_cells_ixs"] = sub
    for direction in ("forward", "backwards"):
        vlm.prepare_markov(sigma_D=2.0, sigma_W=4.0, direction=direction, cells_ixs=sub)
        out[f"tr_subset_{direction}"] = vlm.tr.toarray()
        vlm.run_markov(n_steps=20)
        out[f"diffused_subset_{direction}"] = np.asarray(vlm.diffused).ravel()
    vlm.prepare_markov(sigma_D=2.0, sigma_W=4.0, direction="backwards")      # back to the state golden_pipeline left
    with np.errstate(all="ignore"):
        vlm.calculate_grid_arrows(smooth=0.8, steps=(12, 10), n_neighbors=30, n_jobs=1)
    for k in ("flow_grid", "flow", "flow_norm", "flow_norm_magnitude", "total_p_mass"):
        out[k] = np.asarray(getattr(vlm, k))
    d = dif.Diffusion()
    out["embedding"], out["delta_embedding"] = vlm.embedding, vlm.delta_embedding
    out["tm2_fwd"] = d.compute_transition_matrix2(vlm.embedding, vlm.delta_embedding, sigma=0.7).toarray()
    out["tm2_rev"] = d.compute_transition_matrix2(vlm.embedding, vlm.delta_embedding, sigma=0.7, reverse=True).toarray()
    knn = vlm.embedding_knn.tocoo()
    out["knn_row"], out["knn_col"] = knn.row, knn.col
    with np.errstate(all="ignore"):
        out["tm1_fwd"] = d.compute_transition_matrix(vlm.embedding_knn, vlm.embedding, vlm.delta_embedding, epsilon=0.01).toarray()
        out["tm1_rev"] = d.compute_transition_matrix(vlm.embedding_knn, vlm.embedding, vlm.delta_embedding, epsilon=0.01, reverse=True).toarray()
    v2 = deepcopy(vlm)
    v2.ra = {"Gene": np.arange(v2.S.shape[0])}
    v2.filter_genes_by_phase_portrait(minR2=0.1, min_gamma=0.05, minCorr=0.1)
    out["filter_kept_genes"] = v2.ra["Gene"]
    out["filter_Sx_sz"] = np.ascontiguousarray(v2.Sx_sz)
    out["filter_gammas"] = v2.gammas
    v3 = deepcopy(vlm)
    v3.ra = {"Gene": np.arange(v3.S.shape[0])}
    v3.filter_genes_good_fit(minR=0.2, min_gamma=0.02)
    out["goodfit_kept_genes"] = v3.ra["Gene"]
    save("next", **out)


def golden_preprocess(ana):
    """Callers upstream of the hot path (analysis.py:134-533, 678-932, 1889-1964): cell/gene filters, feature scores,
    size normalisations, PCA and the two deprecated default_* drivers, recorded from the reference on one small dataset."""
    from copy import deepcopy
    rng = np.random.default_rng(20180814)
    G, C = 420, 260
    S, U, t = synth_counts(rng, G, C)
    S[:30] = rng.poisson(0.03, (30, C))                      # barely detected genes
    U[:30] = rng.poisson(0.02, (30, C))
    S[30:40] = rng.poisson(60.0, (10, C))                    # house-keeping-like (mean above max_expr_avg)
    labels = np.array(["tiny" if x > 0.93 else "c%d" % int(x * 4) for x in t])
    colors = {u: [0.1 + 0.15 * i, 0.5, 0.9 - 0.1 * i] for i, u in enumerate(np.unique(labels))}
    class Snapshots(dict):           # the reference updates some matrices in place (U_sz[:, mask] = ...): record copies
        def __setitem__(self, k, v):
            dict.__setitem__(self, k, np.array(v, copy=True))
    out = Snapshots()
    out["S"], out["U"], out["labels"] = S, U, labels
    vlm = make_vlm(ana, S, U)
    vlm.ca["Clusters"] = labels.copy()
    vlm.set_clusters(labels, cluster_colors_dict=colors)
    out["cluster_ix"], out["cluster_uid"], out["colorandum"] = vlm.cluster_ix, vlm.cluster_uid, vlm.colorandum
    # filter_cells (on a copy)
    keep_cells = rng.random(C) > 0.15
    v0 = deepcopy(vlm)
    v0.ts = rng.random((C, 2))
    v0.filter_cells(keep_cells)
    out["keep_cells"] = keep_cells
    out["fc_S"], out["fc_initial_cell_size"], out["fc_cluster_ix"], out["fc_CellID"] = v0.S, v0.initial_cell_size, v0.cluster_ix, v0.ca["CellID"]
    # detection levels
    vlm.score_detection_levels(min_expr_counts=40, min_cells_express=20, min_expr_counts_U=15, min_cells_express_U=10)
    out["detection_level_selected"] = vlm.detection_level_selected
    vlm.filter_genes(by_detection_levels=True)
    out["genes_after_detection"] = vlm.ra["Gene"]
    # cv vs mean, three variants
    vlm.score_cv_vs_mean(N=200, max_expr_avg=40)
    out["cv_mean_score"], out["cv_mean_selected"] = vlm.cv_mean_score, vlm.cv_mean_selected
    v1 = deepcopy(vlm)
    v1.score_cv_vs_mean(N=150, max_expr_avg=40, winsorize=True, winsor_perc=(1, 99.5), svr_gamma=0.4)
    out["cv_mean_score_winsor"], out["cv_mean_selected_winsor"] = v1.cv_mean_score, v1.cv_mean_selected
    v1.score_cv_vs_mean(N=120, max_expr_avg=40, sort_inverse=True, min_expr_cells=5, min_expr_avg=0.05)
    out["cv_mean_score_inverse"], out["cv_mean_selected_inverse"] = v1.cv_mean_score, v1.cv_mean_selected
    vlm.score_cv_vs_mean(N=150, max_expr_avg=30, which="U")
    out["Ucv_mean_score"], out["Ucv_mean_selected"] = vlm.Ucv_mean_score, vlm.Ucv_mean_selected
    vlm.score_cluster_expression(min_avg_U=0.02, min_avg_S=0.08)
    out["U_avgs"], out["S_avgs"], out["clu_avg_selected"] = vlm.U_avgs, vlm.S_avgs, vlm.clu_avg_selected
    vlm.robust_size_factor(pc=0.1, which="both")
    out["size_factor"], out["Usize_factor"] = vlm.size_factor, vlm.Usize_factor
    # custom index-array filter + keep_unfiltered on a copy
    v2 = deepcopy(vlm)
    v2.filter_genes(by_custom_array=np.arange(5, 200, 3), keep_unfiltered=True)
    out["genes_custom_index"] = v2.ra["Gene"]
    out["S_prefilter_sum"] = np.asarray(v2.S_prefilter.sum())
    v2.custom_filter_attributes(["cv_mean_score"], np.isin(np.arange(len(v2.cv_mean_score)), np.arange(5, 200, 3)))
    out["custom_attr_cv_mean_score"] = v2.cv_mean_score
    vlm.filter_genes(by_cv_vs_mean=True, by_cluster_expression=True)
    out["genes_after_cv_cluster"] = vlm.ra["Gene"]
    # normalisations
    va = deepcopy(vlm)
    va.normalize_by_total(min_perc_U=0.5)
    out["nt_small_U_pop"], out["nt_S_sz"], out["nt_U_sz"], out["nt_S_norm"] = va.small_U_pop, va.S_sz, va.U_sz, va.S_norm
    va.adjust_totS_totU(normalize_total=True)
    out["adj_S_sz"], out["adj_U_sz"] = va.S_sz, va.U_sz
    vb = deepcopy(vlm)
    vb.normalize_by_total(min_perc_U=5, skip_low_U_pop=False, same_size_UnS=True)
    out["nt2_S_sz"], out["nt2_U_sz"] = vb.S_sz, vb.U_sz
    vb.adjust_totS_totU(skip_low_U_pop=False, fit_with_low_U=False, normalize_total=False)
    out["adj2_U_sz"] = vb.U_sz
    vc = deepcopy(vlm)
    vc.normalize_by_size_factor(min_perc_U=0.5)
    out["sf_S_sz"], out["sf_U_sz"] = vc.S_sz, vc.U_sz
    # PCA + default_fit_preparation's choices
    va.perform_PCA()
    out["pcs"], out["explained_variance_ratio"], out["pca_components"] = va.pcs, va.pca.explained_variance_ratio_, va.pca.components_
    out["pca_mean"], out["pca_explained_variance"] = va.pca.mean_, va.pca.explained_variance_
    vd = deepcopy(va)
    vd.perform_PCA(n_components=15)
    out["pcs15"] = vd.pcs
    va.knn_imputation(n_pca_dims=8, k=10, balanced=True, b_sight=80, b_maxl=40, n_jobs=1)
    va.normalize_median()
    out["nm_Sx_sz"], out["nm_Ux_sz"] = va.Sx_sz, va.Ux_sz
    ve = deepcopy(va)
    ve.normalize_median(which="imputed", skip_low_U_pop=False)
    out["nm2_Ux_sz"] = ve.Ux_sz
    va.normalize("imputed", size=False, log=True)
    out["Sx_norm"] = va.Sx_norm
    va._perform_PCA_imputed(n_components=6)
    out["pcsx"] = va.pcsx
    # the deprecated one-call drivers
    vf = make_vlm(ana, S, U)
    vf.set_clusters(labels, cluster_colors_dict=colors)
    vf.default_filter_and_norm(min_expr_counts=30, min_cells_express=15, N=180)
    out["dfn_genes"], out["dfn_S_sz"], out["dfn_U_sz"] = vf.ra["Gene"], vf.S_sz, vf.U_sz
    vf.default_fit_preparation(k=12, n_comps=8)
    out["dfp_Sx_sz"], out["dfp_Ux_sz"], out["dfp_pcs"] = vf.Sx_sz, vf.Ux_sz, vf.pcs
    out["dfp_n_comps_rule"] = np.asarray(int(np.where(np.diff(np.diff(np.cumsum(vf.pca.explained_variance_ratio_)) > 0.002))[0][0]))
    save("preprocess", **{k: np.asarray(v) for k, v in out.items()})


CFG1_SEED, CFG1_G, CFG1_C, CFG1_P, CFG1_K = 20180813, 2000, 3000, 20, 30


def cfg1_inputs():
    """BASELINE.json configs[0] stand-in (DentateGyrus.loom is not available offline): 3000 cells x 2000 genes
    of seeded synthetic counts and a 20-d search space derived from the same latent time.  Pure NumPy from a PCG64 seed, so
    tests/test_gpu_fullsize.py regenerates the identical arrays on the GPU box; only the reference's OUTPUTS are stored."""
    rng = np.random.default_rng(CFG1_SEED)
    S, U, t = synth_counts(rng, CFG1_G, CFG1_C)
    pcs = np.concatenate([np.stack([8.0 * t, 3.0 * np.sin(3.0 * t)], 1), rng.normal(0, 0.25, (CFG1_C, CFG1_P - 2))], 1) + rng.normal(0, 0.05, (CFG1_C, CFG1_P))
    return S, U, pcs


def golden_cfg1(ana):
    """fit_gammas() with its defaults (maxmin_diag weights, weighted offset fit by L-BFGS-B; analysis.py:1120-1260,
    estimation.py:212-241, 337-366) at cfg1 size.  Stored: gammas, q, R2 (float32) + checksums of the pooled matrices."""
    S, U, pcs = cfg1_inputs()
    vlm = make_vlm(ana, S, U)
    vlm.normalize("both", size=True, log=True)
    vlm.pcs = pcs
    vlm.knn_imputation(k=CFG1_K, n_pca_dims=CFG1_P, n_jobs=4)
    vlm.fit_gammas()
    save("cfg1", gammas=vlm.gammas, q=vlm.q, R2=vlm.R2, seed=np.asarray(CFG1_SEED), shape=np.asarray([CFG1_G, CFG1_C, CFG1_P, CFG1_K]),
         Sx_sum=np.asarray(vlm.Sx_sz.sum()), Ux_sum=np.asarray(vlm.Ux_sz.sum()), Sx_row17=np.ascontiguousarray(vlm.Sx_sz[17]),
         knn_row0=np.sort(vlm.knn[0].indices))


def golden_steady(ana):
    """fit_gammas(steady_state_bool=<list mask>) - the part of the reference's steady-state handling that runs
    (analysis.py:1159-1162, 1223-1257): the UNWEIGHTED fits on tmpS[:, mask], tmpU[:, mask].  (An ndarray mask raises on
    `if steady_state_bool:`; the weighted fits fail on broadcasting because W is not subset - both checked here.)
    Inputs: Sx, Ux of pipeline.npz; the mask is regenerated from its seed by the tests."""
    g = np.load(os.path.join(HERE, "pipeline.npz"))
    S, U = g["S"], g["U"]
    vlm = make_vlm(ana, S, U)
    vlm.Sx = vlm.Sx_sz = np.array(g["Sx"])
    vlm.Ux = vlm.Ux_sz = np.array(g["Ux"])
    C = S.shape[1]
    mask = np.random.default_rng(20180812).random(C) < 0.6
    out = dict(mask=mask)
    vlm.fit_gammas(steady_state_bool=list(mask), fit_offset=False, weighted=False)
    out["gammas_plain"] = vlm.gammas.copy()
    vlm.fit_gammas(steady_state_bool=list(mask), fit_offset=True, weighted=False)
    out["gammas_offset"], out["q_offset"] = vlm.gammas.copy(), vlm.q.copy()
    vlm.fit_gammas(steady_state_bool=list(mask), fit_offset=False, fixperc_q=True, weighted=False)
    out["gammas_fixq"], out["q_fixq"] = vlm.gammas.copy(), vlm.q.copy()
    for bad, exc in ((dict(steady_state_bool=mask, weighted=False), ValueError), (dict(steady_state_bool=list(mask)), ValueError)):
        try:
            vlm.fit_gammas(**bad)
            raise SystemExit(f"the reference was expected to fail on {list(bad)}")
        except exc as e:
            print("reference raises as documented:", type(e).__name__, str(e)[:90])
    save("steady", **out)


if __name__ == "__main__":
    est, nb, dif, ana = load_reference()
    which = set(sys.argv[1:])
    if not which or "coldeltacor" in which:
        golden_coldeltacor(est)
    if not which or "fits" in which:
        golden_fits(est)
    if not which or "neighbors" in which:
        golden_neighbors(nb)
    if not which or "pipeline" in which:
        golden_pipeline(ana, dif)
    if not which or "preprocess" in which:
        golden_preprocess(ana)
    if not which or "cfg1" in which:
        golden_cfg1(ana)
    if not which or "steady" in which:
        golden_steady(ana)
