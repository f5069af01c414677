"""GPU parity tests for the callers upstream of the hot path (velocyto_amd.preprocess.PreprocessMixin): the same call
sequence the reference ran for tests/golden/preprocess.npz (tests/golden/make_golden.py::golden_preprocess), through
the device facade, against the recorded outputs - plus gene_stats against numpy on ragged shapes.

Masks / index sets are compared exactly; matrices to rounding (f64 storage) or f32 tolerances (f32 storage).  PCA: the
leading components (well-separated eigenvalues) to 1e-7 absolute; the explained-variance spectrum as a whole.
"""
from copy import deepcopy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

TOL = {"float64": (1e-11, 1e-11), "float32": (3e-5, 3e-5)}


@pytest.fixture(scope="module")
def vcy():
    import velocyto_amd
    from velocyto_amd import ops
    ops.require_gpu()
    return velocyto_amd


def close(a, b, rtol, atol):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def start(vcy, g, dtype):
    vlm = vcy.analysis.VelocytoLoom.from_arrays(g["S"], g["U"], dtype=dtype)
    labels = g["labels"]
    colors = {u: [0.1 + 0.15 * i, 0.5, 0.9 - 0.1 * i] for i, u in enumerate(np.unique(labels))}
    vlm.set_clusters(labels, cluster_colors_dict=colors)
    return vlm, colors


@pytest.mark.parametrize("shape", [(1, 1), (7, 3), (300, 257), (1000, 70)])
@pytest.mark.parametrize("kind", ["float32", "float64", "uint16", "uint8"])
def test_gene_stats_matches_numpy(vcy, shape, kind):
    from velocyto_amd import ops
    C, G = shape
    rng = np.random.default_rng(C * 131 + G)
    X = rng.poisson(1.5, (G, C)).astype(np.float64)               # (genes, cells)
    if kind in ("uint16", "uint8"):
        M = ops.CountMatrix.from_genes_major(X.astype(np.uint16), narrow=kind == "uint8")
        assert M.t.dtype == (torch.uint8 if kind == "uint8" else torch.int16)
    else:
        M = ops.CellMatrix.from_genes_major(X, getattr(torch, kind))
    st = ops.gene_stats(M).cpu().numpy()
    np.testing.assert_array_equal(st[0], X.sum(1))
    np.testing.assert_array_equal(st[1], (X * X).sum(1))
    np.testing.assert_array_equal(st[2], (X > 0).sum(1))
    np.testing.assert_array_equal(st[3], X.max(1))
    scale, mask = rng.random(C) + 0.5, rng.random(C) > 0.3
    lo, hi = np.full(G, 0.4), 1.0 + rng.random(G) * 2
    if not mask.any():
        mask[0] = True
    st = ops.gene_stats(M, cell_scale=scale, lo=lo, hi=hi, cell_mask=mask).cpu().numpy()
    Y = np.clip(X * scale[None, :], lo[:, None], hi[:, None])[:, mask]
    np.testing.assert_allclose(st[0], Y.sum(1), rtol=1e-13)
    np.testing.assert_allclose(st[1], (Y * Y).sum(1), rtol=1e-13)
    np.testing.assert_array_equal(st[2], (Y > 0).sum(1))
    np.testing.assert_allclose(st[3], Y.max(1), rtol=1e-15)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_filters_and_scores(vcy, golden, dtype):
    g = golden("preprocess")
    vlm, _ = start(vcy, g, dtype)
    assert np.array_equal(vlm.cluster_ix, g["cluster_ix"]) and np.array_equal(vlm.cluster_uid, g["cluster_uid"])
    np.testing.assert_allclose(vlm.colorandum, g["colorandum"])
    # filter_cells on a copy
    v0 = deepcopy(vlm)
    v0.ts = np.zeros((len(g["keep_cells"]), 2))
    v0.filter_cells(g["keep_cells"])
    assert np.array_equal(v0.S, g["fc_S"]) and np.array_equal(v0.ca["CellID"], g["fc_CellID"])
    np.testing.assert_allclose(v0.initial_cell_size, g["fc_initial_cell_size"])
    assert np.array_equal(v0.cluster_ix, g["fc_cluster_ix"]) and v0.ts.shape[0] == g["keep_cells"].sum()
    assert v0.U.shape == v0.S.shape == v0.A.shape if hasattr(v0, "A") else True
    # detection levels -> gene filter
    vlm.score_detection_levels(min_expr_counts=40, min_cells_express=20, min_expr_counts_U=15, min_cells_express_U=10)
    assert np.array_equal(vlm.detection_level_selected, g["detection_level_selected"])
    vlm.filter_genes(by_detection_levels=True)
    assert np.array_equal(vlm.ra["Gene"], g["genes_after_detection"])
    assert np.array_equal(vlm.S, g["S"][g["genes_after_detection"]].astype(float))
    # CV vs mean: libsvm stops at a dual tolerance of 1e-3, so inputs that differ in the last bit (summation order) move the
    # fitted curve by up to ~1e-3; scores are compared to 5e-3 and the rank-based selections may swap genes at the cut
    def same_set(a, b, n):
        assert a.shape == b.shape and (a != b).sum() <= max(2, int(0.02 * n)), f"{(a != b).sum()} genes differ"
    vlm.score_cv_vs_mean(N=200, max_expr_avg=40)
    close(vlm.cv_mean_score, g["cv_mean_score"], 0, 5e-3)
    same_set(vlm.cv_mean_selected, g["cv_mean_selected"], 200)
    v1 = deepcopy(vlm)
    v1.score_cv_vs_mean(N=150, max_expr_avg=40, winsorize=True, winsor_perc=(1, 99.5), svr_gamma=0.4)
    close(v1.cv_mean_score, g["cv_mean_score_winsor"], 0, 5e-3)
    same_set(v1.cv_mean_selected, g["cv_mean_selected_winsor"], 150)
    v1.score_cv_vs_mean(N=120, max_expr_avg=40, sort_inverse=True, min_expr_cells=5, min_expr_avg=0.05)
    close(v1.cv_mean_score, g["cv_mean_score_inverse"], 0, 5e-3)
    same_set(v1.cv_mean_selected, g["cv_mean_selected_inverse"], 120)
    vlm.score_cv_vs_mean(N=150, max_expr_avg=30, which="U")
    close(vlm.Ucv_mean_score, g["Ucv_mean_score"], 0, 5e-3)
    same_set(vlm.Ucv_mean_selected, g["Ucv_mean_selected"], 150)
    vlm.cv_mean_selected, vlm.Ucv_mean_selected = g["cv_mean_selected"], g["Ucv_mean_selected"]     # continue from the recorded masks
    vlm.score_cluster_expression(min_avg_U=0.02, min_avg_S=0.08)
    close(vlm.U_avgs, g["U_avgs"], 1e-13, 0)
    close(vlm.S_avgs, g["S_avgs"], 1e-13, 0)
    assert np.array_equal(vlm.clu_avg_selected, g["clu_avg_selected"])
    vlm.robust_size_factor(pc=0.1, which="both")
    close(vlm.size_factor, g["size_factor"], 1e-6 if dtype == "float32" else 1e-12, 0)
    close(vlm.Usize_factor, g["Usize_factor"], 1e-6 if dtype == "float32" else 1e-12, 0)
    v2 = deepcopy(vlm)
    v2.filter_genes(by_custom_array=np.arange(5, 200, 3), keep_unfiltered=True)
    assert np.array_equal(v2.ra["Gene"], g["genes_custom_index"])
    assert float(v2.S_prefilter.sum()) == float(g["S_prefilter_sum"])
    v2.custom_filter_attributes(["cv_mean_score"], np.isin(np.arange(len(v2.cv_mean_score)), np.arange(5, 200, 3)))
    close(v2.cv_mean_score, g["custom_attr_cv_mean_score"], 0, 5e-3)
    vlm.filter_genes(by_cv_vs_mean=True, by_cluster_expression=True)
    assert np.array_equal(vlm.ra["Gene"], g["genes_after_cv_cluster"])
    with pytest.raises(AssertionError):
        vlm.filter_genes()
    with pytest.raises(NotImplementedError):
        vlm.gene_knn_imputation()


def filtered(vcy, g, dtype):
    vlm, _ = start(vcy, g, dtype)
    keep = np.zeros(g["S"].shape[0], dtype=bool)
    keep[g["genes_after_cv_cluster"]] = True
    vlm.filter_genes(by_custom_array=keep)
    vlm.size_factor = g["size_factor"]
    return vlm


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_size_normalisations(vcy, golden, dtype):
    g = golden("preprocess")
    rt, at = TOL[dtype]
    va = filtered(vcy, g, dtype)
    va.normalize_by_total(min_perc_U=0.5)
    assert np.array_equal(va.small_U_pop, g["nt_small_U_pop"])
    close(va.S_sz, g["nt_S_sz"], rt, at)
    close(va.U_sz, g["nt_U_sz"], rt, at)
    close(va.S_norm, g["nt_S_norm"], rt, at)
    va.adjust_totS_totU(normalize_total=True)
    close(va.S_sz, g["adj_S_sz"], max(rt, 1e-10), at)
    close(va.U_sz, g["adj_U_sz"], max(rt, 2e-6), at)             # SVR prediction in the factor: two SMO solvers stopped at tol=1e-3 (libsvm there, csrc/svr.hip here; tests/test_gpu_svr.py)
    vb = filtered(vcy, g, dtype)
    vb.normalize_by_total(min_perc_U=5, skip_low_U_pop=False, same_size_UnS=True)
    close(vb.S_sz, g["nt2_S_sz"], rt, at)
    close(vb.U_sz, g["nt2_U_sz"], rt, at)
    vb.adjust_totS_totU(skip_low_U_pop=False, fit_with_low_U=False, normalize_total=False)
    close(vb.U_sz, g["adj2_U_sz"], max(rt, 2e-6), at)
    vc = filtered(vcy, g, dtype)
    vc.normalize_by_size_factor(min_perc_U=0.5)
    close(vc.S_sz, g["sf_S_sz"], rt, at)
    close(vc.U_sz, g["sf_U_sz"], rt, at)
    vd = filtered(vcy, g, dtype)
    vd.initial_Ucell_size = np.ones_like(vd.initial_Ucell_size)
    with pytest.raises(ValueError):
        vd.normalize_by_total()
    # the pooling fast path must see the rescaled U_sz (factor * counts with the adjusted factor)
    va.pcs = g["pcs"]
    va.knn_imputation(n_pca_dims=8, k=10, balanced=True, b_sight=80, b_maxl=40, n_jobs=1)
    assert va.dev("Sx_sz") is va.dev("Sx") and va.dev("Ux_sz") is va.dev("Ux")      # one matrix under both names after the pooling ...
    sx_before = va.Sx.copy()
    va.normalize_median()
    assert va.dev("Sx_sz") is not va.dev("Sx") and np.array_equal(va.Sx, sx_before)   # ... until one of them is rescaled: Sx keeps its values
    close(va.Sx_sz, g["nm_Sx_sz"], max(rt, 1e-9), max(at, 1e-9))
    close(va.Ux_sz, g["nm_Ux_sz"], max(rt, 2e-6), max(at, 1e-9))                 # pooled from the SVR-adjusted U_sz
    ve = deepcopy(va)
    ve.normalize_median(which="imputed", skip_low_U_pop=False)
    close(ve.Ux_sz, g["nm2_Ux_sz"], max(rt, 2e-6), max(at, 1e-9))


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_pca_on_device(vcy, golden, dtype):
    g = golden("preprocess")
    va = filtered(vcy, g, dtype)
    va.normalize_by_total(min_perc_U=0.5)
    va.perform_PCA()
    f32 = dtype == "float32"
    k = 40
    assert va.pcs.shape == g["pcs"].shape and va.pca.components_.shape == g["pca_components"].shape
    # (the covariance product is csrc/gram.hip on the f64 matrix cores: explained variances to 1e-9 of scikit-learn's in f64 storage)
    close(va.pca.explained_variance_ratio_, g["explained_variance_ratio"], 2e-5 if f32 else 1e-9, 1e-9 if f32 else 1e-13)
    close(va.pca.explained_variance_, g["pca_explained_variance"], 2e-5 if f32 else 1e-9, 1e-9 if f32 else 1e-13)
    close(va.pca.mean_, g["pca_mean"], 1e-6 if f32 else 1e-12, 1e-6 if f32 else 1e-12)
    close(va.pcs[:, :k], g["pcs"][:, :k], 0, 2e-3 if f32 else 1e-7)
    close(va.pca.components_[:k], g["pca_components"][:k], 0, 2e-4 if f32 else 1e-7)
    # scores are orthogonal with the recorded variances, whatever the trailing directions are
    gram = va.pcs.T @ va.pcs / (va.pcs.shape[0] - 1)
    close(np.diag(gram), va.pca.explained_variance_, 1e-6, 1e-9)
    assert np.abs(gram - np.diag(np.diag(gram))).max() < (1e-4 if f32 else 1e-9)
    vd = deepcopy(va)
    vd.perform_PCA(n_components=15)
    close(vd.pcs, g["pcs15"], 0, 2e-3 if f32 else 1e-7)
    n_rule = int(np.where(np.diff(np.diff(np.cumsum(va.pca.explained_variance_ratio_)) > 0.002))[0][0])
    va.Sx_norm = g["Sx_norm"]
    va._perform_PCA_imputed(n_components=6)
    close(va.pcsx, g["pcsx"], 0, 2e-3 if f32 else 1e-7)
    with pytest.raises(ValueError):
        va.perform_PCA(n_components=10**6)
    # more genes than cells: dual (cells x cells) route against the oracle-independent identity pcs @ components = centred data
    X = g["nt_S_norm"][:, :50]
    vs = vcy.analysis.VelocytoLoom.from_arrays(g["S"][g["genes_after_cv_cluster"]][:, :50], g["U"][g["genes_after_cv_cluster"]][:, :50], dtype=dtype)
    vs.S_norm = X
    vs.perform_PCA(n_components=20)
    Xc = X.T - X.T.mean(0)
    Uu, s, Vt = np.linalg.svd(Xc, full_matrices=False)
    close(np.abs(vs.pcs), np.abs(Uu[:, :20] * s[:20]), 0, 2e-3 if f32 else 1e-7)
    assert n_rule >= 0


@pytest.mark.parametrize("dtype", ["float64"])
def test_default_drivers(vcy, golden, dtype):
    g = golden("preprocess")
    vf, _ = start(vcy, g, dtype)
    vf.default_filter_and_norm(min_expr_counts=30, min_cells_express=15, N=180)
    assert np.array_equal(vf.ra["Gene"], g["dfn_genes"])
    close(vf.S_sz, g["dfn_S_sz"], 1e-10, 1e-11)
    close(vf.U_sz, g["dfn_U_sz"], 2e-6, 1e-11)                # through adjust_totS_totU's SVR
    vf.default_fit_preparation(k=12, n_comps=8)
    close(vf.pcs[:, :8], g["dfp_pcs"][:, :8], 0, 1e-7)
    close(vf.Sx_sz, g["dfp_Sx_sz"], 1e-8, 1e-9)
    close(vf.Ux_sz, g["dfp_Ux_sz"], 2e-6, 1e-9)
    assert int(np.where(np.diff(np.diff(np.cumsum(vf.pca.explained_variance_ratio_)) > 0.002))[0][0]) == int(g["dfp_n_comps_rule"])


def test_pca_subspace_iteration_matches_the_exact_route():
    """perform_PCA on wide input: the converged subspace iteration that replaces the O(G^3) covariance
    eigensolver when few components of a large matrix are asked for (scikit-learn's own `auto` rule switches to a
    randomised solver there) against the exact route on the same matrix."""
    import velocyto_amd
    from velocyto_amd import ops
    from velocyto_amd.preprocess import DevicePCA
    dev = ops.require_gpu()
    gen = torch.Generator(device=dev).manual_seed(5)
    C, G, r = 6000, 5000, 12
    # low-rank signal with a decaying spectrum + noise: a gap behind the leading components, like real expression data
    L = torch.randn((C, r), generator=gen, device=dev, dtype=torch.float64) * torch.linspace(12, 3, r, device=dev, dtype=torch.float64)
    X = L @ torch.randn((r, G), generator=gen, device=dev, dtype=torch.float64) + torch.randn((C, G), generator=gen, device=dev, dtype=torch.float64) + 2.0
    M = ops.CellMatrix.from_cells_major(X, torch.float64)
    exact = DevicePCA(n_components=10, svd_solver="full")
    p_exact = exact.fit_transform(M)
    auto = DevicePCA(n_components=10)                       # min(C, G) > 4096 and 10 << 5000: takes the subspace route
    p_auto = auto.fit_transform(M)
    assert hasattr(auto, "n_iter_") and not hasattr(exact, "n_iter_") and auto.n_iter_ < 60
    np.testing.assert_allclose(auto.explained_variance_, exact.explained_variance_, rtol=1e-8)
    np.testing.assert_allclose(auto.explained_variance_ratio_, exact.explained_variance_ratio_, rtol=1e-8)
    np.testing.assert_allclose(np.abs(np.sum(auto.components_ * exact.components_, 1)), 1.0, atol=1e-8)      # same directions
    np.testing.assert_allclose(auto.components_, exact.components_, atol=1e-6)                                  # same signs (svd_flip)
    np.testing.assert_allclose(p_auto, p_exact, atol=1e-5 * np.abs(p_exact).max())
    np.testing.assert_allclose(auto.mean_, exact.mean_, rtol=1e-13)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("C,G", [(37, 5), (300, 130), (1000, 257), (5000, 700), (4200, 1153)])
def test_gram_kernel_against_the_library_gemm(dtype, C, G):
    """vcy_gram (csrc/gram.hip, v_mfma_f64_16x16x4_f64): the centred covariance product of perform_PCA (analysis.py:678-702)
    against torch's fp64 GEMM of the explicitly centred matrix - "bit-near": both accumulate in fp64, in different orders.
    Shapes cover one partial tile, ragged tile edges, odd gene counts (scalar tail of a column pair), cell counts that are not
    a multiple of the slab, and the split of the cells over several workgroups per tile (partials reduced in a fixed order)."""
    import velocyto_amd
    from velocyto_amd import ops
    dev = ops.require_gpu()
    gen = torch.Generator(device=dev).manual_seed(C + G)
    X = torch.randn((C, G), generator=gen, device=dev, dtype=torch.float64) * torch.linspace(0.5, 3.0, G, device=dev, dtype=torch.float64) + \
        torch.linspace(-2.0, 40.0, G, device=dev, dtype=torch.float64)                   # means far from zero: centring matters
    M = ops.CellMatrix.from_cells_major(X, getattr(torch, dtype))
    Xs = M.t[:, :G].double()                                                             # what the kernel reads (f32 storage rounds)
    mean = ops.col_means(M)
    assert float((mean - Xs.mean(0)).abs().max()) <= 1e-12 * 40
    A = Xs - mean
    ref = A.T @ A
    got = ops.gram(M, mean)
    scale = float(ref.diagonal().max())
    assert got.shape == (G, G) and float((got - ref).abs().max()) <= 1e-12 * scale
    assert torch.equal(got, got.T), "both triangles must hold the same bits"
    assert torch.equal(got, ops.gram(M, mean)), "fixed summation order: run-to-run identical"
    raw = ops.gram(M, None)                                                              # no centring: X^T X
    assert float((raw - Xs.T @ Xs).abs().max()) <= 1e-12 * float((Xs * Xs).sum(0).max())
    # the thin block product of the subspace iteration, with an ASYMMETRIC right-hand side (a row <-> column slip cannot hide)
    thin = {}
    for L in (1, 7, 50, 64, 130):
        Y = torch.randn((C, L), generator=gen, device=dev, dtype=torch.float64) * torch.arange(1, L + 1, device=dev, dtype=torch.float64)
        w = ops.gram_tn(M, mean, Y)
        wref = A.T @ Y
        assert w.shape == (G, L) and float((w - wref).abs().max()) <= 1e-12 * float(wref.abs().max()) + 1e-9
        thin[L] = (Y, w)
    # the tile columns past the last gene are staged with the rest of the slab row (whatever lies in the row's padding): they must only
    # ever reach output rows / columns that are not written
    if M.ld > G:
        M.t[:, G:] = float("nan")
        assert torch.equal(ops.gram(M, mean), got) and torch.equal(ops.gram(M, None), raw)
        for L, (Y, w) in thin.items():
            assert torch.equal(ops.gram_tn(M, mean, Y), w)


def test_gram_kernel_identity_probe():
    """Lane maps of the f64 MFMA pinned with an identity probe: X = [I; 0] (cells x genes) against an asymmetric Y gives
    X^T Y = Y's first rows exactly - any slip in the A / B / D lane layout (the f64 D map differs from the f32 one) moves entries."""
    import velocyto_amd
    from velocyto_amd import ops
    dev = ops.require_gpu()
    G, C, L = 200, 333, 96
    X = torch.zeros((C, G), dtype=torch.float64, device=dev)
    X[:G] = torch.eye(G, dtype=torch.float64, device=dev)
    Y = (torch.arange(C, device=dev, dtype=torch.float64)[:, None] * 1000.0 + torch.arange(L, device=dev, dtype=torch.float64)[None, :]).contiguous()
    M = ops.CellMatrix.from_cells_major(X, torch.float64)
    assert torch.equal(ops.gram_tn(M, None, Y), Y[:G])
    assert torch.equal(ops.gram(M, None), torch.eye(G, dtype=torch.float64, device=dev))


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("C,G,N", [(5, 3, 1), (130, 17, 7), (300, 65, 64), (1000, 257, 50), (900, 515, 100), (700, 1153, 130), (257, 96, 257)])
def test_gemm_nt_kernel_against_the_library_gemm(dtype, C, G, N):
    """vcy_gemm_nt (csrc/gram.hip, v_mfma_f64_16x16x4_f64, slabs DMA'd into LDS): the products of perform_PCA that contract over the
    genes (analysis.py:678-702 through sklearn's PCA: projection on a thin block, `transform`, the cells' Gram matrix), against torch's
    fp64 GEMM of the explicitly centred matrix.  Shapes cover one partial tile, ragged tile edges in both directions, gene counts that
    end inside a slab (zero padding), thin (N = 1) and wide blocks; f32 storage of X beside an fp64 block (256-byte slab rows)."""
    import velocyto_amd
    from velocyto_amd import ops
    dev = ops.require_gpu()
    gen = torch.Generator(device=dev).manual_seed(C * 7 + G + N)
    X = torch.randn((C, G), generator=gen, device=dev, dtype=torch.float64) * torch.linspace(0.5, 3.0, G, device=dev, dtype=torch.float64) + \
        torch.linspace(-2.0, 40.0, G, device=dev, dtype=torch.float64)
    M = ops.CellMatrix.from_cells_major(X, getattr(torch, dtype))
    Xs = M.t[:, :G].double()
    mean = ops.col_means(M)
    A = Xs - mean
    B = torch.randn((N, G), generator=gen, device=dev, dtype=torch.float64) * torch.arange(1, N + 1, device=dev, dtype=torch.float64)[:, None]
    # plain product, asymmetric right-hand side
    got = ops.gemm_nt(M, B)
    ref = Xs @ B.T
    assert got.shape == (C, N) and float((got - ref).abs().max()) <= 1e-12 * float(ref.abs().max()) + 1e-12
    assert torch.equal(got, ops.gemm_nt(M, B)), "fixed summation order: run-to-run identical"
    # the centred projection (X - m) B^T through the column correction - what DevicePCA's subspace pass and `transform` call
    proj = ops.gemm_nt(M, B, col_corr=B @ mean)
    pref = A @ B.T
    assert float((proj - pref).abs().max()) <= 1e-11 * float(ref.abs().max()) + 1e-12       # (error of the expansion: relative to the uncentred product)
    # into a strided output (a column block of a wider buffer), as the subspace pass writes Y
    wide = torch.full((C, N + 3), 7.0, dtype=torch.float64, device=dev)
    ops.gemm_nt(M, B, col_corr=B @ mean, out=wide[:, :N])
    assert torch.equal(wide[:, :N], proj) and bool((wide[:, N:] == 7.0).all())
    # the cells' own Gram matrix, centred by algebra: (X - m)(X - m)^T = X X^T - a 1^T - 1 a^T + m.m, a = X m
    a = ops.gemm_nt(M, mean[None, :])[:, 0].contiguous()
    assert float((a - Xs @ mean).abs().max()) <= 1e-12 * float((Xs @ mean).abs().max()) + 1e-12
    gram = ops.gemm_nt(M, M, row_corr=a, col_corr=a, c0=float(mean @ mean))
    gref = A @ A.T
    assert float((gram - gref).abs().max()) <= 1e-11 * float((Xs @ Xs.T).abs().max())
    assert float((gram - gram.T).abs().max()) <= 1e-11 * float((Xs @ Xs.T).abs().max())


def test_gemm_nt_kernel_identity_probe():
    """Lane maps and the piece swizzle of vcy_gemm_nt pinned exactly: X = [I | 0] (cells x genes) against an asymmetric B gives X B^T =
    B^T's first rows, bit for bit - in both storage types of X (128- and 256-byte slab rows of B)."""
    import velocyto_amd
    from velocyto_amd import ops
    dev = ops.require_gpu()
    C, G, N = 200, 333, 96
    X = torch.zeros((C, G), dtype=torch.float64, device=dev)
    X[:, :C] = torch.eye(C, dtype=torch.float64, device=dev)
    B = (torch.arange(N, device=dev, dtype=torch.float64)[:, None] * 1000.0 + torch.arange(G, device=dev, dtype=torch.float64)[None, :]).contiguous()
    for dt in (torch.float64, torch.float32):
        M = ops.CellMatrix.from_cells_major(X, dt)
        assert torch.equal(ops.gemm_nt(M, B), B[:, :C].T.contiguous())


def test_pca_products_are_the_librarys_own_kernels(monkeypatch):
    """perform_PCA's device route sends no (cells x genes) operand through a library GEMM: with torch.matmul made to refuse operands of the
    matrix's size, all three routes (covariance, dual for more genes than cells, subspace iteration) still run - and agree."""
    import velocyto_amd
    from velocyto_amd import ops
    from velocyto_amd.preprocess import DevicePCA
    dev = ops.require_gpu()
    gen = torch.Generator(device=dev).manual_seed(11)
    C, G, r = 5000, 4500, 8
    L = torch.randn((C, r), generator=gen, device=dev, dtype=torch.float64) * torch.linspace(10, 3, r, device=dev, dtype=torch.float64)
    X = L @ torch.randn((r, G), generator=gen, device=dev, dtype=torch.float64) + torch.randn((C, G), generator=gen, device=dev, dtype=torch.float64) + 1.5
    M = ops.CellMatrix.from_cells_major(X, torch.float64)
    Mw = ops.CellMatrix.from_cells_major(X[:900], torch.float64)          # 900 cells x 4500 genes: the dual route
    real_matmul = torch.Tensor.__matmul__

    def guarded(a, b):
        for t in (a, b):
            assert not (torch.is_tensor(t) and t.dim() == 2 and min(t.shape) >= 900 and max(t.shape) >= 4000), f"library GEMM on a {tuple(t.shape)} operand"
        return real_matmul(a, b)
    monkeypatch.setattr(torch.Tensor, "__matmul__", guarded)
    exact = DevicePCA(n_components=6, svd_solver="full")
    p_exact = exact.fit_transform(M)
    sub = DevicePCA(n_components=6, svd_solver="subspace")
    p_sub = sub.fit_transform(M)
    dual = DevicePCA(n_components=6, svd_solver="full")
    p_dual = dual.fit_transform(Mw)
    monkeypatch.undo()
    np.testing.assert_allclose(sub.explained_variance_, exact.explained_variance_, rtol=1e-8)
    np.testing.assert_allclose(p_sub, p_exact, atol=1e-5 * np.abs(p_exact).max())
    Xw = X[:900].cpu().numpy()
    Uu, s, Vt = np.linalg.svd(Xw - Xw.mean(0), full_matrices=False)
    np.testing.assert_allclose(np.abs(p_dual), np.abs(Uu[:, :6] * s[:6]), atol=1e-8 * s[0])
    np.testing.assert_allclose(dual.singular_values_, s[:6], rtol=1e-10)
