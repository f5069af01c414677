"""Callers upstream of the hot path: the filtering / scoring / size-normalisation / PCA methods of
``velocyto.analysis.VelocytoLoom`` (analysis.py:134-533, 678-932, 1441-1450, 1889-1964), same names, arguments
and attributes, mixed into the device-resident facade.

What runs where:
  * every reduction over the (genes x cells) matrices is a HIP kernel: per-gene sum / sum of squares / number of
    expressing cells / winsorised moments (``vcy_gene_stats``), per-gene percentiles (``vcy_gene_quantiles``), per-cell
    totals (``vcy_row_sums``), scaling and log (``vcy_scale_log``);
  * gene / cell subsetting and the per-cell rescalings are index / broadcast plumbing on the device tensors;
  * PCA is the covariance route on the device in fp64, identical to scikit-learn's ``PCA`` up to rounding, with scikit-learn's
    sign convention: every product with a (cells x genes) operand is a hand-written f64 matrix-core kernel of csrc/gram.hip (vcy_gram,
    vcy_gram_tn, vcy_gemm_nt - the matrix is read as stored, centring is algebra), the small eigensolvers / QR are library calls;
  * the SVR noise models of score_cv_vs_mean (one point per gene) and adjust_totS_totU (one point per cell), which the
    reference delegates to scikit-learn / libsvm, are fitted by the same SMO iteration on the device (``DeviceSVR`` ->
    ``vcy_svr_rbf_fit``: 0.5-0.7 s at 50 000 points against about a minute for libsvm); t-SNE stays with scikit-learn.
Plotting is out of scope (``plot=True`` is accepted and ignored).
"""
from __future__ import annotations

import logging
from copy import deepcopy
from typing import Any, Dict, List, Tuple

import numpy as np
import torch
from scipy import sparse

from . import ops
from .ops import CellMatrix


def colormap_fun(x: np.ndarray) -> np.ndarray:
    """analysis.py:2385-2389: 20 colours, tab20b even entries followed by tab20c odd entries."""
    import matplotlib
    tab20b, tab20c = matplotlib.colormaps["tab20b"], matplotlib.colormaps["tab20c"]
    colors20 = np.vstack((tab20b(np.linspace(0., 1, 20))[::2], tab20c(np.linspace(0, 1, 20))[1::2]))
    return colors20[np.mod(x, 20)]


class DeviceSVR:
    """``sklearn.svm.SVR(kernel="rbf")`` on one scalar feature, the only way the reference uses it (analysis.py:280-282, 324-326,
    844-851), fitted on the device by libsvm's own iteration (csrc/svr.hip).  Same constructor arguments, ``fit`` /
    ``predict`` and the fitted attributes ``support_``, ``support_vectors_``, ``dual_coef_``, ``intercept_``, ``n_iter_``,
    ``fit_status_``.  Agrees with libsvm to its stopping tolerance (``tol``, in units of the target)."""

    def __init__(self, kernel: str = "rbf", gamma: Any = "scale", C: float = 1.0, epsilon: float = 0.1, tol: float = 1e-3, max_iter: int = -1):
        if kernel != "rbf":
            raise NotImplementedError("DeviceSVR covers the RBF kernel only (what the reference fits)")
        self.kernel, self.gamma, self.C, self.epsilon, self.tol, self.max_iter = kernel, gamma, C, epsilon, tol, max_iter

    @staticmethod
    def _column(X) -> np.ndarray:
        X = np.asarray(X, dtype=np.float64)
        if X.ndim == 2 and X.shape[1] == 1:
            return X[:, 0]
        raise ValueError(f"Expected a 2D array with a single feature, got shape {X.shape}")       # sklearn refuses 1D input as well

    def fit(self, X, y) -> "DeviceSVR":
        x = self._column(X)
        y = np.asarray(y, dtype=np.float64).ravel()
        if len(x) != len(y):
            raise ValueError(f"Found input variables with inconsistent numbers of samples: [{len(x)}, {len(y)}]")
        if self.gamma == "scale":
            var = x.var()
            self._gamma = 1.0 / var if var != 0 else 1.0
        elif self.gamma == "auto":
            self._gamma = 1.0
        else:
            self._gamma = float(self.gamma)
        coef, intercept, info = ops.svr_fit(x, y, C=self.C, epsilon=self.epsilon, gamma=self._gamma, tol=self.tol, max_iter=self.max_iter)
        info = info.cpu().numpy()
        if info[2]:
            raise RuntimeError("svr_rbf_fit: the workgroups of the solver lost each other at a grid barrier (device busy with another "
                               "context?); set VCY_SVR_WG=1 to run the fit on a single workgroup")
        self.n_iter_, self.fit_status_ = int(info[0]), int(not info[1])
        if not info[1]:
            logging.warning(f"Solver terminated early (max_iter={self.max_iter}). Consider pre-processing your data with StandardScaler or MinMaxScaler.")
        self._x, self._coef, self._intercept = torch.as_tensor(x, device=coef.device), coef, intercept
        c = coef.cpu().numpy()
        self.support_ = np.flatnonzero(c).astype(np.int32)
        self.support_vectors_ = x[self.support_, None]
        self.dual_coef_ = c[None, self.support_]
        self.intercept_ = intercept.cpu().numpy()
        return self

    def predict(self, X) -> np.ndarray:
        return ops.svr_predict(self._x, self._coef, self._intercept, self._column(X), self._gamma).cpu().numpy()


class DevicePCA:
    """What ``sklearn.decomposition.PCA`` leaves behind after ``fit`` (the attributes the reference and its tutorial
    read: ``explained_variance_ratio_``, ``components_``, ...), computed on the device."""

    def __init__(self, n_components=None, svd_solver: str = "auto", tol: float = 1e-9, max_iter: int = 60, random_state: int = 0):
        self.n_components, self.svd_solver, self.tol, self.max_iter, self.random_state = n_components, svd_solver, tol, max_iter, random_state

    def fit_transform(self, X: CellMatrix, block: int = 8192) -> np.ndarray:
        """X: cells-major device matrix (samples = cells).  Returns pcs (C, n_components) float64."""
        C, G = X.C, X.G
        dev = X.t.device
        k = min(C, G) if self.n_components is None else int(self.n_components)
        if not 1 <= k <= min(C, G):
            raise ValueError(f"n_components={self.n_components!r} must be between 1 and min(n_samples, n_features)={min(C, G)}")
        # scikit-learn's PCA(svd_solver="auto") - what the reference gets (analysis.py:698) - switches to a randomised solver
        # when few components of a large matrix are asked for; the exact covariance route below is O(min(C, G)^3) in its
        # eigensolver and holds a min(C, G)^2 fp64 matrix (7.2 GB at 30 000 unfiltered genes).  Same switch here, with a
        # solver that is iterated to convergence instead of stopped after a fixed number of passes.
        if self.svd_solver == "subspace" or (self.svd_solver == "auto" and min(C, G) > 4096 and k <= 0.25 * min(C, G)):
            return self._fit_subspace(X, k, block)
        mean = ops.col_means(X)
        if G <= C:      # covariance of the genes (G x G): the centred Gram product on the f64 matrix cores (csrc/gram.hip), then the
            #             symmetric eigenproblem of the small matrix as a library call (rocSOLVER through torch.linalg.eigh)
            gram = ops.gram(X, mean)
            w, V = torch.linalg.eigh(gram)
            w, V = w.flip(0).clamp_(min=0.0), V.flip(1)                  # descending
            comps = V[:, :k].T.contiguous()                               # (k, G)
        else:           # fewer cells than genes: eigenvectors of the (C x C) Gram matrix of the cells, mapped back.  Both products on the
            #             f64 matrix cores, X read as stored: (X - m)(X - m)^T = X X^T - a 1^T - 1 a^T + m.m with a = X m (vcy_gemm_nt), and
            #             the map back (X - m)^T U_k is the contraction over the cells of the subspace pass (vcy_gram_tn)
            a = ops.gemm_nt(X, mean[None, :])[:, 0].contiguous()
            w, Uc = torch.linalg.eigh(ops.gemm_nt(X, X, row_corr=a, col_corr=a, c0=float(mean @ mean)))
            w, Uc = w.flip(0).clamp_(min=0.0), Uc.flip(1)
            comps = ops.gram_tn(X, mean, Uc[:, :k].contiguous()).T / torch.sqrt(w[:k]).clamp(min=1e-300)[:, None]
        # sklearn.utils.extmath.svd_flip(u_based_decision=False): the largest-|.| loading of every component is positive
        idx = comps.abs().argmax(1)
        comps = comps * torch.sign(comps[torch.arange(k, device=dev), idx])[:, None]
        pcs = ops.gemm_nt(X, comps, col_corr=(comps * mean).sum(1))     # (X - mean) comps^T = X comps^T - 1 (comps mean)
        total_var = float(w.sum()) / (C - 1)
        self.components_ = comps.cpu().numpy()
        self.explained_variance_ = (w[:k] / (C - 1)).cpu().numpy()
        self.explained_variance_ratio_ = self.explained_variance_ / total_var
        self.singular_values_ = np.sqrt(w[:k].cpu().numpy())
        self.mean_ = mean.cpu().numpy()
        self.n_components_, self.n_samples_, self.n_features_in_ = k, C, G
        self.noise_variance_ = float(w[k:].sum() / (C - 1) / max(1, min(C, G) - k)) if k < min(C, G) else 0.0
        self._pcs_dev = pcs
        return pcs.cpu().numpy()

    def _fit_subspace(self, X: CellMatrix, k: int, block: int) -> np.ndarray:
        """Leading k principal components by blocked subspace iteration on the centred matrix A = X - mean (never formed):
        Z <- orth(A^T (A Z)) on a G x (k + 20) block until the Ritz values stop moving (relative `tol`), then Rayleigh-Ritz.
        Every pass is two streams over X on the f64 matrix cores - the projection over the genes (vcy_gemm_nt) and the contraction
        over the cells (vcy_gram_tn), X read as stored both times; the small (k + 20)-wide QR / eigenproblems are library calls;
        memory O((C + G) (k + 20)).  Deterministic (seeded start), converged rather than truncated, so the leading
        components agree with the exact route to the tolerance whenever the spectrum has a gap behind them."""
        C, G = X.C, X.G
        dev = X.t.device
        l = min(min(C, G), k + 20)
        st = ops.gene_stats(X)                                                  # per-gene sum of squares over the cells (vcy_gene_stats, fp64)
        mean = ops.col_means(X)
        total_var = float((st[1].sum() - C * (mean * mean).sum()) / (C - 1))    # trace of the covariance
        ldy = l + (l % 2)                                                       # (vcy_gram_tn wants 16-byte aligned rows of Y)
        Ybuf = torch.zeros((C, ldy), dtype=torch.float64, device=dev)

        def AtA(Z):                                                             # A^T (A Z), A centred
            # Y = A Z (C x l) = X Z - 1 (mean Z): the thin projection over the genes (vcy_gemm_nt), X read as stored; A^T Y (G x l): the
            # contraction over the cells with the centring folded into the fragment read (vcy_gram_tn) - both on the f64 matrix cores
            # (the block transposed to rows over the genes by the layout's own transpose kernel; mean Z as a column sum - the library's
            #  gemv takes 4.9 ms for this 30 000 x 50 product, more than the projection itself)
            ops.gemm_nt(X, CellMatrix.from_genes_major(Z, torch.float64), col_corr=(Z * mean[:, None]).sum(0), out=Ybuf[:, :Z.shape[1]])
            return ops.gram_tn(X, mean, Ybuf) if ldy == Z.shape[1] else ops.gram_tn(X, mean, Ybuf)[:, :Z.shape[1]]
        gen = torch.Generator(device=dev).manual_seed(int(self.random_state))
        Z = torch.linalg.qr(torch.randn((G, l), generator=gen, device=dev, dtype=torch.float64))[0]
        prev = None
        self.converged_ = False
        for it in range(int(self.max_iter)):
            W = AtA(Z)
            ritz = torch.linalg.eigvalsh(Z.T @ W).flip(0)[:k]
            Z = torch.linalg.qr(W)[0]
            if prev is not None and float(((ritz - prev).abs() / ritz.abs().clamp(min=1e-300)).max()) < self.tol:
                self.converged_ = True
                break
            prev = ritz
        self.n_iter_ = it + 1
        if not self.converged_:
            import warnings
            warnings.warn(f"DevicePCA: subspace iteration stopped at max_iter={self.max_iter} before the leading {k} Ritz values settled to "
                          f"rel tol {self.tol:g} (clustered spectrum?); components_ may be unconverged - raise max_iter or use the exact route "
                          "(svd_solver='full')", RuntimeWarning, stacklevel=2)
        W = AtA(Z)
        w, V = torch.linalg.eigh(Z.T @ W)
        w, V = w.flip(0).clamp_(min=0.0), V.flip(1)
        comps = (Z @ V[:, :k]).T.contiguous()                                   # (k, G)
        idx = comps.abs().argmax(1)
        comps = comps * torch.sign(comps[torch.arange(k, device=dev), idx])[:, None]     # sklearn's svd_flip
        pcs = ops.gemm_nt(X, comps, col_corr=(comps * mean).sum(1))
        self.components_ = comps.cpu().numpy()
        self.explained_variance_ = (w[:k] / (C - 1)).cpu().numpy()
        self.explained_variance_ratio_ = self.explained_variance_ / total_var
        self.singular_values_ = np.sqrt(w[:k].cpu().numpy())
        self.mean_ = mean.cpu().numpy()
        self.n_components_, self.n_samples_, self.n_features_in_ = k, C, G
        self.noise_variance_ = float((total_var - self.explained_variance_.sum()) / max(1, min(C, G) - k))
        self._pcs_dev = pcs
        return pcs.cpu().numpy()

    def __getstate__(self):
        return {k: v for k, v in self.__dict__.items() if k != "_pcs_dev"}      # checkpoints carry host arrays only

    def transform(self, Xcg: np.ndarray) -> np.ndarray:
        """(samples, genes) host array -> scores (host convenience, as sklearn)."""
        return (np.asarray(Xcg, dtype=np.float64) - self.mean_) @ self.components_.T


class PreprocessMixin:
    """Methods of VelocytoLoom upstream of knn_imputation (mixed into analysis.VelocytoLoom)."""

    # ------------------------------------------------------------------ subsetting
    def _subset(self, names, keep, axis: str) -> None:
        """Row (cells) or column (genes) subset of device matrices, count layers included."""
        sel = ops.select_cells if axis == "cells" else ops.select_genes
        keep_t = torch.as_tensor(np.asarray(keep), device=ops.require_gpu())
        counts = self.__dict__.setdefault("_counts", {})
        for name in names:
            if name not in self._dev:
                continue
            cnt = counts.get(name)
            self._set_dev(name, sel(self._dev[name], keep_t))
            if cnt is not None:
                counts[name] = sel(cnt, keep_t)
            if axis == "cells":
                sc = self.__dict__.get("_sz_scale", {})
                if name + "_sz" in sc:
                    sc.pop(name + "_sz")

    def filter_cells(self, bool_array: np.ndarray) -> None:
        """analysis.py:134-163: keep the cells where bool_array is True (S, U, A and the per-cell annotations)."""
        bool_array = np.asarray(bool_array)
        self._subset(("S", "U", "A"), bool_array, "cells")
        self.initial_cell_size = self.initial_cell_size[bool_array]
        self.initial_Ucell_size = self.initial_Ucell_size[bool_array]
        for attr in ("ts", "size_factor"):
            try:
                setattr(self, attr, getattr(self, attr)[bool_array])
            except Exception:
                pass
        self.ca = {k: v[bool_array] for k, v in self.ca.items()}
        try:
            self.cluster_labels = self.cluster_labels[bool_array]
            self.colorandum = self.colorandum[bool_array, :]
        except AttributeError:
            pass

    # which score attribute stands behind each switch of filter_genes (set by score_cluster_expression, score_cv_vs_mean,
    # score_detection_levels respectively)
    _GENE_FILTERS = (("by_cluster_expression", "clu_avg_selected"), ("by_cv_vs_mean", "cv_mean_selected"),
                     ("by_detection_levels", "detection_level_selected"))

    def filter_genes(self, by_detection_levels: bool = False, by_cluster_expression: bool = False, by_cv_vs_mean: bool = False,
                     by_custom_array: Any = None, keep_unfiltered: bool = False) -> None:
        """analysis.py:441-497: S, U and ra are cut down to the genes that pass every requested filter - the conjunction of the
        stored score masks that were asked for and of a custom mask (boolean) or gene-number list (integer)."""
        asked = dict(by_detection_levels=by_detection_levels, by_cluster_expression=by_cluster_expression, by_cv_vs_mean=by_cv_vs_mean)
        custom = by_custom_array if type(by_custom_array) is np.ndarray else None
        assert any(asked.values()) or custom is not None, "At least one of the filtering methods needs to be True"
        n_genes = self.dev("S").G
        masks = []
        for switch, attr in self._GENE_FILTERS:
            if asked[switch]:
                assert hasattr(self, attr), f"{attr} was not found"
                masks.append(np.asarray(getattr(self, attr), dtype=bool))
        if custom is not None:
            if custom.dtype == bool:
                masks.append(custom)
            elif custom.dtype == int:              # gene numbers: everything not listed goes
                listed = np.zeros(n_genes, dtype=bool)
                listed[custom[(custom >= 0) & (custom < n_genes)]] = True
                masks.append(listed)
        keep = np.logical_and.reduce([np.ones(n_genes, dtype=bool)] + masks)
        if keep_unfiltered:
            if hasattr(self, "U_prefilter"):
                logging.debug("Attributes *_prefilter are already present and were overwritten")
            for name in ("U", "S"):
                setattr(self, name + "_prefilter", sparse.csr_matrix(getattr(self, name)))
            self.ra_prefilter = deepcopy(self.ra)
        self._subset(("U", "S"), keep, "genes")
        self.ra = {key: values[keep] for key, values in self.ra.items()}

    def custom_filter_attributes(self, attr_names: List[str], bool_filter: np.ndarray) -> None:
        """analysis.py:535-571 (numbering of the reference file: the block before _normalize_S)."""
        for attr in attr_names:
            transpose_flag = attr[-2:] == ".T"
            obj = getattr(self, attr[:-2] if transpose_flag else attr)
            if type(obj) is dict:
                setattr(self, attr, {k: v[bool_filter] for k, v in obj.items()})
            elif type(obj) is np.ndarray:
                if len(obj.shape) > 1:
                    setattr(self, attr, obj[..., bool_filter] if transpose_flag else obj[bool_filter, :])
                else:
                    setattr(self, attr, obj[bool_filter])
            else:
                raise NotImplementedError(f"The filtering of an object of type {type(obj)} is not defined")

    # ------------------------------------------------------------------ clusters
    def set_clusters(self, cluster_labels: np.ndarray, cluster_colors_dict: Dict[str, List[float]] = None, colormap: Any = None) -> None:
        """analysis.py:165-199."""
        self.cluster_labels = np.array(cluster_labels)
        if self.cluster_labels.dtype == "O":
            self.cluster_labels = self.cluster_labels.astype(np.bytes_)
        if cluster_colors_dict:
            self.colorandum = np.array([cluster_colors_dict[i] for i in cluster_labels])
            self.cluster_colors_dict = cluster_colors_dict
            self.colormap = None
        else:
            fun = colormap_fun if colormap is None else colormap
            self.colormap = colormap
            self.colorandum = fun(self.cluster_ix)
            cluster_uid = self.cluster_uid
            self.cluster_colors_dict = {cluster_uid[i]: fun(i) for i in range(len(cluster_uid))}

    @property
    def cluster_uid(self) -> np.ndarray:
        return np.unique(self.cluster_labels)

    @property
    def cluster_ix(self) -> np.ndarray:
        return np.unique(self.cluster_labels, return_inverse=True)[1]

    # ------------------------------------------------------------------ gene scores
    def _layer_for_stats(self, name: str):
        """The uint16 count layer when the matrix still is one (2-byte reads), else the float matrix."""
        return self.__dict__.get("_counts", {}).get(name, self.dev(name))

    def score_detection_levels(self, min_expr_counts: int = 50, min_cells_express: int = 20, min_expr_counts_U: int = 0,
                               min_cells_express_U: int = 0) -> None:
        """analysis.py:456-475: detection_level_selected from per-gene totals and numbers of expressing cells."""
        sS = ops.gene_stats(self._layer_for_stats("S")).cpu().numpy()
        sU = ops.gene_stats(self._layer_for_stats("U")).cpu().numpy()
        self.detection_level_selected = ((sS[0] >= min_expr_counts) & (sS[2] >= min_cells_express) &
                                         (sU[0] >= min_expr_counts_U) & (sU[2] >= min_cells_express_U))

    def score_cv_vs_mean(self, N: int = 3000, min_expr_cells: int = 2, max_expr_avg: float = 20, min_expr_avg: int = 0, svr_gamma: float = None,
                         winsorize: bool = False, winsor_perc: Tuple[float, float] = (1, 99.5), sort_inverse: bool = False, which: str = "S",
                         plot: bool = False) -> None:
        """analysis.py:201-345: CV-vs-mean noise model (SVR on log2 mean -> log2 CV) and the N genes most above it.
        Per-gene detection, mean and std(ddof=1) - optionally of the values winsorised to per-gene percentiles - come from
        one or two streaming passes on the device; the SVR (G points) is fitted on the device too (DeviceSVR)."""
        name = "S" if which == "S" else "U"
        M = self._layer_for_stats(name)
        C, G = M.C, M.G
        if winsorize:
            if min_expr_cells <= ((100 - winsor_perc[1]) * C * 0.01):
                min_expr_cells = int(np.ceil((100 - winsor_perc[1]) * G * 0.01)) + 2          # sic: genes, as the reference (:248)
                logging.debug(f"min_expr_cells is too low for winsorization with upper_perc ={winsor_perc[1]}, upgrading to min_expr_cells ={min_expr_cells}")
        st = ops.gene_stats(M).cpu().numpy()
        mean_all = st[0] / C
        detected_bool = (st[2] > min_expr_cells) & (mean_all < max_expr_avg) & (mean_all > min_expr_avg)
        if winsorize:
            q = ops.gene_quantiles(self.dev(name), list(winsor_perc))
            st = ops.gene_stats(M, lo=q[0], hi=q[1]).cpu().numpy()
        mu = (st[0] / C)[detected_bool]
        var = (st[1][detected_bool] - C * mu * mu) / (C - 1)
        sigma = np.sqrt(np.maximum(var, 0.0))
        with np.errstate(divide="ignore", invalid="ignore"):
            cv = sigma / mu
            log_m, log_cv = np.log2(mu), np.log2(cv)
        if svr_gamma is None:
            svr_gamma = 150. / len(mu)
            logging.debug(f"svr_gamma set to {svr_gamma}")
        clf = DeviceSVR(gamma=svr_gamma)
        clf.fit(log_m[:, None], log_cv)
        score = log_cv - clf.predict(log_m[:, None])
        if sort_inverse:
            score = -score
        nth_score = np.sort(score)[::-1][N]
        full = np.zeros(detected_bool.shape)
        full[~detected_bool] = np.min(score) - 1e-16
        full[detected_bool] = score
        if which == "S":
            self.cv_mean_score, self.cv_mean_selected = full, full >= nth_score
        else:
            self.Ucv_mean_score, self.Ucv_mean_selected = full, full >= nth_score

    def score_cluster_expression(self, min_avg_U: float = 0.02, min_avg_S: float = 0.08) -> None:
        """analysis.py:441-454 + estimation.clusters_stats (estimation.py:369-389): per-cluster gene averages (clusters of
        at most 40 cells report the overall average), masked streaming passes on the device."""
        from .estimation import clusters_stats
        self.U_avgs, self.S_avgs = clusters_stats(self.dev("U"), self.dev("S"), self.cluster_uid, self.cluster_ix, size_limit=40)
        self.clu_avg_selected = (self.U_avgs.max(1) > min_avg_U) & (self.S_avgs.max(1) > min_avg_S)

    def robust_size_factor(self, pc: float = 0.1, which: str = "both") -> None:
        """analysis.py:347-439: per-cell median over the selected genes of 2**(log2(x + pc) - gene mean of log2), mean 1."""
        def one(name, selected):
            sub = ops.select_genes(self.dev(name), torch.as_tensor(selected))
            Y = torch.log2(sub.t[:, :sub.G].double() + pc)
            R = torch.exp2(Y - Y.mean(0, keepdim=True))
            srt = torch.sort(R, dim=1).values
            n = sub.G
            med = 0.5 * (srt[:, (n - 1) // 2] + srt[:, n // 2])          # numpy's median
            med = med.cpu().numpy()
            return med / np.mean(med)
        if which in ("both", "S"):
            self.size_factor = one("S", self.cv_mean_selected)
        if which in ("both", "U"):
            self.Usize_factor = one("U", self.Ucv_mean_selected)

    # ------------------------------------------------------------------ size normalisations
    def _norm_common(self, cell_size, Ucell_size, S_relative, min_perc_U, skip_low_U_pop, same_size_UnS):
        target_cell_size = np.median(cell_size)
        min_Ucell_size = np.percentile(Ucell_size, min_perc_U)
        if min_Ucell_size < 2:
            raise ValueError(f"min_perc_U={min_perc_U} corresponds to total Unspliced of 1 molecule of less. Please choose higher value or filter our these cell")
        self.small_U_pop = Ucell_size < min_Ucell_size
        target_Ucell_size = target_cell_size if same_size_UnS else np.median(Ucell_size[~self.small_U_pop])
        self._normalize_S(relative_size=S_relative, target_size=target_cell_size)
        if skip_low_U_pop:
            self._normalize_U(relative_size=np.clip(self.initial_Ucell_size, min_Ucell_size, None), target_size=target_Ucell_size)
        else:
            self._normalize_U(relative_size=self.initial_Ucell_size, target_size=target_Ucell_size)

    def normalize_by_total(self, min_perc_U: float = 0.5, plot: bool = False, skip_low_U_pop: bool = True, same_size_UnS: bool = False) -> None:
        """analysis.py:704-758."""
        self._norm_common(self.initial_cell_size, self.initial_Ucell_size, self.initial_cell_size, min_perc_U, skip_low_U_pop, same_size_UnS)

    def normalize_by_size_factor(self, min_perc_U: float = 0.5, plot: bool = False, skip_low_U_pop: bool = True, same_size_UnS: bool = False) -> None:
        """analysis.py:760-818 (cell sizes of the CURRENT S / U; S is divided by ``size_factor``)."""
        cell_size = ops.row_sums(self.dev("S")).cpu().numpy()
        Ucell_size = ops.row_sums(self.dev("U")).cpu().numpy()
        self._norm_common(cell_size, Ucell_size, self.size_factor, min_perc_U, skip_low_U_pop, same_size_UnS)

    def _scale_cells(self, name: str, factor: np.ndarray) -> None:
        """M[:, c] *= factor[c] on the device (in a new matrix when `name` is new)."""
        f = torch.as_tensor(np.asarray(factor, dtype=np.float64), device=self.dev(name).t.device)
        M = self.dev(name)
        if any(k != name and v is M for k, v in self._dev.items()):      # shared with another attribute (Sx_sz is Sx after knn_imputation)
            M = CellMatrix(M.t.clone(), M.G)
            self._dev[name] = M
        M.t.mul_(f[:, None].to(M.dtype))
        self._host.pop(name, None)
        sc = self.__dict__.get("_sz_scale", {})
        if name in sc:
            sc[name] = sc[name] * f

    def adjust_totS_totU(self, skip_low_U_pop: bool = True, normalize_total: bool = False, fit_with_low_U: bool = True, svr_C: float = 100,
                         svr_gamma: float = 1e-6, plot: bool = False) -> None:
        """analysis.py:820-868: SVR of total U_sz on total S_sz per cell; U_sz is rescaled towards the prediction."""
        svr = DeviceSVR(C=svr_C, kernel="rbf", gamma=svr_gamma)
        X, y = ops.row_sums(self.dev("S_sz")).cpu().numpy(), ops.row_sums(self.dev("U_sz")).cpu().numpy()
        if fit_with_low_U:
            svr.fit(X[:, None], y)
            predicted = svr.predict(X[:, None])
        else:
            svr.fit(X[~self.small_U_pop, None], y[~self.small_U_pop])
            predicted = np.copy(y)
            predicted[~self.small_U_pop] = svr.predict(X[~self.small_U_pop, None])
        with np.errstate(divide="ignore", invalid="ignore"):
            adj_factor = predicted / y
        adj_factor[~np.isfinite(adj_factor)] = 1
        if skip_low_U_pop:
            adj_factor = np.where(~self.small_U_pop, adj_factor, 1.0)
        self._scale_cells("U_sz", adj_factor)
        if normalize_total:
            self.normalize_median(which="renormalize", skip_low_U_pop=skip_low_U_pop)

    def normalize_median(self, which: str = "imputed", skip_low_U_pop: bool = True) -> None:
        """analysis.py:870-905: every cell rescaled to the median cell total (raw size-normalised or imputed matrices)."""
        C = self.dev("U_sz").C if which == "renormalize" else self.dev("Ux").C
        if not hasattr(self, "small_U_pop") and skip_low_U_pop:
            self.small_U_pop = np.zeros(C, dtype=bool)
            logging.warning("object does not have the attribute `small_U_pop`, so all the unspliced will be normalized by relative size, this might cause the overinflation the unspliced counts of cells where only few unspliced molecules were detected")

        def factors(tot, subset):
            f = np.ones(len(tot))
            with np.errstate(divide="ignore", invalid="ignore"):
                f[subset] = np.median(tot[subset]) / tot[subset]
            return f
        everyone = np.ones(C, dtype=bool)
        if which == "renormalize":
            self._scale_cells("S_sz", factors(ops.row_sums(self.dev("S_sz")).cpu().numpy(), everyone))
            self._scale_cells("U_sz", factors(ops.row_sums(self.dev("U_sz")).cpu().numpy(), ~self.small_U_pop if skip_low_U_pop else everyone))
        elif which == "imputed":
            self._set_dev("Sx_sz", self.dev("Sx").clone())
            self._scale_cells("Sx_sz", factors(ops.row_sums(self.dev("Sx")).cpu().numpy(), everyone))
            self._set_dev("Ux_sz", self.dev("Ux").clone())
            self._scale_cells("Ux_sz", factors(ops.row_sums(self.dev("Ux")).cpu().numpy(), ~self.small_U_pop if skip_low_U_pop else everyone))

    # ------------------------------------------------------------------ PCA / t-SNE
    def _pca_input(self, which: str, div_by_std: bool) -> CellMatrix:
        X = self.dev(which)
        if div_by_std:
            # the reference writes X.T / X.std(0), which only broadcasts when cells == genes; what it can mean for a
            # (genes, cells) matrix is each cell divided by its own standard deviation over the genes
            t = X.t[:, :X.G].double()
            X = CellMatrix.from_cells_major(t / t.std(1, unbiased=False, keepdim=True), torch.float64)
        return X

    def perform_PCA(self, which: str = "S_norm", n_components: int = None, div_by_std: bool = False) -> None:
        """analysis.py:678-702: PCA with cells as samples -> ``pca`` (fitted attributes) and ``pcs`` (cells, npcs)."""
        self.pca = DevicePCA(n_components=n_components)
        self.pcs = self.pca.fit_transform(self._pca_input(which, div_by_std))

    def _perform_PCA_imputed(self, n_components: int = None) -> None:
        """analysis.py:918-920."""
        self.pcax = DevicePCA(n_components=n_components)
        self.pcsx = self.pcax.fit_transform(self.dev("Sx_norm"))

    def perform_TSNE(self, n_dims: int = 2, perplexity: float = 30, initial_pos: np.ndarray = None, theta: float = 0.5, n_pca_dim: int = None,
                     max_iter: int = 1000) -> None:
        """analysis.py:1441-1450: Barnes-Hut t-SNE of the leading PCs (scikit-learn, as the reference; its ``n_iter`` keyword is
        ``max_iter`` since scikit-learn 1.5)."""
        import inspect
        from sklearn.manifold import TSNE
        kw = "max_iter" if "max_iter" in inspect.signature(TSNE.__init__).parameters else "n_iter"
        bh_tsne = TSNE(n_components=n_dims, perplexity=perplexity, angle=theta, init="random" if initial_pos is None else initial_pos, **{kw: max_iter})
        self.ts = bh_tsne.fit_transform(self.pcs[:, :n_pca_dim])

    # ------------------------------------------------------------------ deprecated one-call drivers
    @staticmethod
    def _default_thresholds(n_cells: int) -> Dict[str, float]:
        """The cell-count heuristics of the deprecated one-call drivers (analysis.py:1909-1918, 1959-1960), in one place."""
        clamp = lambda lo, x, hi: max(lo, min(hi, x))
        return {"min_expr_counts": clamp(20, n_cells * 2.25e-3, 100), "min_cells_express": clamp(10, n_cells * 1.5e-3, 50),
                "N": clamp(1000, int((n_cells / 1000) ** (1 / 3) / 0.0008), 5000), "min_avg_U": 0.01, "min_avg_S": 0.08,
                "k": int(clamp(10, np.ceil(n_cells * 0.02), 1000))}

    def default_filter_and_norm(self, min_expr_counts: int = None, min_cells_express: int = None, N: int = None, min_avg_U: float = None,
                                min_avg_S: float = None) -> None:
        """analysis.py:1889-1940: detection filter -> CV-vs-mean feature selection -> detection filter on the unspliced layer
        (and per-cluster expression when clusters are set) -> total-count normalisations."""
        logging.warning("DEPRECATION WARNING - the current function is deprecated. Please refer to documentation for default parameters usage")
        given = dict(min_expr_counts=min_expr_counts, min_cells_express=min_cells_express, N=N, min_avg_U=min_avg_U, min_avg_S=min_avg_S)
        t = {**self._default_thresholds(self.dev("S").C), **{name: v for name, v in given.items() if v is not None}}
        for layer in ("S", "U"):                   # (only for the initial cell sizes; the normalised values are recomputed at the end)
            self.normalize(layer, size=True, log=False)
        self.score_detection_levels(min_expr_counts=t["min_expr_counts"], min_cells_express=t["min_cells_express"])
        self.filter_genes(by_detection_levels=True)
        self.score_cv_vs_mean(N=t["N"], max_expr_avg=40)
        self.filter_genes(by_cv_vs_mean=True)
        half = lambda x: int(x / 2) + 1
        self.score_detection_levels(min_expr_counts=0, min_cells_express=0, min_expr_counts_U=half(t["min_expr_counts"]),
                                    min_cells_express_U=half(t["min_cells_express"]))
        clustered = hasattr(self, "cluster_labels")
        if clustered:
            self.score_cluster_expression(min_avg_U=t["min_avg_U"], min_avg_S=t["min_avg_S"])
        self.filter_genes(by_detection_levels=True, by_cluster_expression=clustered)
        self.normalize_by_total()
        self.adjust_totS_totU(normalize_total=True)

    def default_fit_preparation(self, k: int = None, n_comps: int = None) -> None:
        """analysis.py:1942-1964: PCA, the number of components from the elbow of the explained variance, balanced kNN pooling with
        sight 8 k and in-degree cap 4 k, median normalisation."""
        logging.warning("DEPRECATION WARNING - the current function is deprecated. Please refer to documetation for default parameters usage")
        n_cells = self.dev("S").C
        self.perform_PCA()
        if n_comps is None:
            gain = np.diff(np.cumsum(self.pca.explained_variance_ratio_))
            n_comps = int(np.flatnonzero(np.diff(gain > 0.002))[0])       # first component where the 0.2 % gain test flips
        k = self._default_thresholds(n_cells)["k"] if k is None else k
        cap = n_cells - 1
        self.knn_imputation(n_pca_dims=n_comps, k=k, balanced=True, b_sight=int(min(k * 8, cap)), b_maxl=int(min(k * 4, cap)))
        self.normalize_median()

    def gene_knn_imputation(self, *args, **kwargs) -> None:
        """analysis.py:1055-1118 smooths genes with the CELL graph's connectivity (``self.knn``), which only
        has a meaning when cells == genes; not reproduced."""
        raise NotImplementedError("gene_knn_imputation: the reference builds the gene weights from the cell kNN graph (analysis.py:1107); "
                                  "there is no well-defined behaviour to mirror")
