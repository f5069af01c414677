"""Drop-in for the parts of ``velocyto/neighbors.py`` that ``VelocytoLoom`` uses
(SURVEY.md section 2 row 3): ``knn_distance_matrix``, ``BalancedKNN``, ``knn_balance`` and its two
greedy loops, ``connectivity_to_weights``, ``convolve_by_sparse_weights``.

The exact kNN search (sklearn ``NearestNeighbors`` in the reference) is the HIP kernel
``k_knn_search``; the greedy balancing loop (numba in the reference) is C++ on the host
(``vcy_balance_knn_host``); pooling is ``k_knn_pool``.  Sparse-matrix *containers* stay scipy, as in
the reference - they carry O(C*k) indices, not data-parallel work.
"""
from __future__ import annotations

import logging
from typing import Any, Optional, Tuple

import numpy as np
import torch
from scipy import sparse

from . import ops
from .ops import CellMatrix

__all__ = ["knn_distance_matrix", "BalancedKNN", "knn_balance", "balance_knn_loop", "balance_knn_loop_constrained",
           "connectivity_to_weights", "convolve_by_sparse_weights", "make_mutual", "min_n", "take_top", "knn_smooth_weights"]


def _search_space(data: np.ndarray, metric: Optional[str]) -> Tuple[np.ndarray, bool]:
    """Euclidean search space for a metric.  "correlation" = 1 - Pearson r: rows are centred and
    scaled to unit norm, then |a-b|^2 = 2 (1 - r) orders identically (distance = |a-b|^2 / 2)."""
    X = np.ascontiguousarray(data, dtype=np.float64)
    if metric == "correlation":
        X = X - X.mean(1, keepdims=True)
        X = X / np.linalg.norm(X, axis=1, keepdims=True)
        return X, True
    if metric not in (None, "euclidean", "minkowski", "l2"):
        raise NotImplementedError(f"metric={metric!r}: only euclidean and correlation are implemented")
    return X, False


def _kneighbors(data: np.ndarray, k: int, metric: Optional[str], include_self: bool) -> Tuple[np.ndarray, np.ndarray]:
    X, corr = _search_space(data, metric)
    idx, dist = ops.knn_search(X, k, include_self=include_self)
    idx, dist = idx.cpu().numpy().astype(np.int64), dist.cpu().numpy()
    if corr:
        dist = dist * dist / 2.0
    return dist, idx


def _kneighbors_device(data: np.ndarray, k: int, metric: Optional[str]) -> Tuple[np.ndarray, np.ndarray, bool]:
    """The kNN lists of knn_distance_matrix (query excluded) with every row sorted by cell number: (indices int32 (n, k),
    distances float64 (n, k), all distances > 0)."""
    import torch
    X, corr = _search_space(data, "correlation" if metric == "correlation" else None)
    idx, dist = ops.knn_search(X, k, include_self=False)
    if corr:
        dist = dist * dist / 2.0
    idx_s, order = torch.sort(idx, dim=1)
    dist_s = torch.gather(dist, 1, order)
    positive = bool((dist_s > 0).all())
    return idx_s.cpu().numpy(), dist_s.cpu().numpy(), positive


def _kneighbors_rows_device(data: np.ndarray, k: int, metric: Optional[str]):
    """_kneighbors_device without the download: (indices int32 (n, k) sorted by cell number, their distances float64 (n, k),
    all distances > 0) with the two matrices still on the device (one scalar comes back)."""
    X, corr = _search_space(data, "correlation" if metric == "correlation" else None)
    idx, dist = ops.knn_search(X, k, include_self=False)
    if corr:
        dist = dist * dist / 2.0
    idx_s, order = torch.sort(idx, dim=1)
    dist_s = torch.gather(dist, 1, order)
    return idx_s.contiguous(), dist_s.contiguous(), bool((dist_s > 0).all())


def knn_distance_matrix(data: np.ndarray, metric: str = None, k: int = 40, mode: str = "connectivity", n_jobs: int = 4
                        ) -> sparse.csr_matrix:
    """neighbors.py:363-376: kNN graph (query excluded), k entries per row, nearest first.
    (The reference ignores `metric` unless it is "correlation", neighbors.py:369-376 - same here.)"""
    dist, idx = _kneighbors(data, k, "correlation" if metric == "correlation" else None, include_self=False)
    n = idx.shape[0]
    vals = dist.ravel() if mode == "distance" else np.ones(n * k)
    return sparse.csr_matrix((vals, idx.ravel(), np.arange(0, n * k + 1, k)), shape=(n, n))


def weights_from_sorted_knn(idx_sorted: np.ndarray, diag: float) -> sparse.csr_matrix:
    """What knn_imputation builds from a kNN graph whose distances are all positive (analysis.py:1006-1010):
        connectivity = (knn > 0).astype(float); connectivity.setdiag(diag); connectivity_to_weights(connectivity)
    written out for rows given as neighbour lists SORTED by cell number, query excluded (idx_sorted: (cells, k)): every row gets k
    ones and `diag` on the diagonal, scaled by the reciprocal of (k + diag) - the same values, the same sorted CSR, without four
    passes of scipy's structure-changing operations over the graph."""
    idx_sorted = np.asarray(idx_sorted)
    n, k = idx_sorted.shape
    me = np.arange(n, dtype=idx_sorted.dtype)[:, None]
    pos = (idx_sorted < me).sum(1)                                   # where the diagonal goes in the sorted row
    cols = np.empty((n, k + 1), dtype=np.int32)
    vals = np.ones((n, k + 1), dtype=np.float64)
    slot = np.arange(k + 1)[None, :]
    before = slot < pos[:, None]
    cols[before] = idx_sorted[before[:, :k]] if k else 0
    after = slot > pos[:, None]
    cols[after] = idx_sorted[after[:, 1:]] if k else 0
    rows = np.arange(n)
    cols[rows, pos] = rows
    vals[rows, pos] = diag
    # scipy's row sum runs over a row in STORED order, and setdiag stores the new diagonal entry after the k ones (the matrix is
    # only sorted later, by the multiply): (1 + ... + 1) + diag, the ones summing exactly
    rowsum = np.float64(k) + np.float64(diag)
    vals = vals * (1.0 / rowsum)
    w = sparse.csr_matrix((vals.ravel(), cols.ravel(), np.arange(0, n * (k + 1) + 1, k + 1)), shape=(n, n))
    w.has_sorted_indices = True
    return w


def balance_knn_loop(dsi: np.ndarray, dist: np.ndarray, lsi: np.ndarray, maxl: int, k: int, return_distance: bool) -> Tuple:
    """neighbors.py:11-72 (numba in the reference, C++ here)."""
    return ops.balance_knn_host(dsi, dist if return_distance else None, lsi, None, maxl, k)


def balance_knn_loop_constrained(dsi: np.ndarray, dist: np.ndarray, lsi: np.ndarray, groups: np.ndarray, maxl: int, k: int,
                                 return_distance: bool) -> Tuple:
    """neighbors.py:75-140."""
    return ops.balance_knn_host(dsi, dist if return_distance else None, lsi, groups, maxl, k)


def knn_balance(dsi: np.ndarray, dist: np.ndarray = None, maxl: int = 200, k: int = 60, constraint: np.ndarray = None
                ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """neighbors.py:143-183: in-degree l = bincount(dsi), processing order = reverse stable argsort of l."""
    dsi = np.ascontiguousarray(dsi, dtype=np.int64)
    l = np.bincount(dsi.flat[:], minlength=dsi.shape[0])
    lsi = np.argsort(l, kind="mergesort")[::-1]
    groups = None if constraint is None else np.asarray(constraint).astype("int64")
    return ops.balance_knn_host(dsi, dist, lsi, groups, maxl, k)


class BalancedKNN:
    """neighbors.py:186-357: greedy in-degree-capped kNN graph with a scikit-learn-like API."""

    def __init__(self, k: int = 50, sight_k: int = 100, maxl: int = 200, constraint: np.ndarray = None, mode: str = "distance",
                 metric: str = "euclidean", n_jobs: int = 4) -> None:
        self.k, self.sight_k, self.maxl, self.mode, self.metric, self.n_jobs = k, sight_k, maxl, mode, metric, n_jobs
        self.dist_new = self.dsi_new = self.l = None
        self.bknn = None
        self.constraint = constraint

    @property
    def n_samples(self) -> int:
        return self.data.shape[0]

    def __getattr__(self, name):
        # `dsi` / `dist`: the reference keeps the full sight lists as host arrays (neighbors.py:282-283); here they stay on the
        # device and become numpy arrays only if somebody reads them
        if name in ("dsi", "dist") and "_lists_dev" in self.__dict__:
            idx, dist = self.__dict__["_lists_dev"]
            val = idx.cpu().numpy().astype(np.int64) if name == "dsi" else dist.cpu().numpy()
            self.__dict__[name] = val
            return val
        raise AttributeError(name)

    def fit(self, data: np.ndarray, sight_k: int = None) -> Any:
        """neighbors.py:226-244 (the search itself runs in kneighbors; there is no index to build)."""
        self.data = data
        self.fitdata = data
        if sight_k is not None:
            self.sight_k = sight_k
        return self

    def kneighbors(self, X: np.ndarray = None, maxl: int = None, mode: str = "distance") -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """neighbors.py:246-289: sight graph of sight_k+1 neighbours (query included, as sklearn's
        kneighbors(data) returns it), then knn_balance."""
        if X is not None:
            self.data = X
        if maxl is not None:
            self.maxl = maxl
        sight = min(int(self.sight_k) + 1, self.fitdata.shape[0])
        logging.debug(f"First search the {self.sight_k} nearest neighbours for {self.n_samples}")
        if self.data is not self.fitdata and not np.array_equal(self.data, self.fitdata):
            # query points other than the fitted ones (neighbors.py:282 with X given): sight lists among the FITTED points
            Xf, corr = _search_space(self.fitdata, self.metric)
            Xq, _ = _search_space(self.data, self.metric)
            idx, dist = ops.knn_query(Xf, Xq, sight)
            self.dsi, self.dist = idx.cpu().numpy().astype(np.int64), dist.cpu().numpy()
            if corr:
                self.dist = self.dist * self.dist / 2.0
        else:
            # the fitted points are their own queries (the path knn_imputation takes; its default sight is the WHOLE dataset,
            # analysis.py:985-988): the (C, sight) lists stay on the device, the greedy loop reads an int32 host copy of the
            # indices only, the distances of the selected entries are gathered on the device (ops.balance_knn_device_lists) -
            # 10 GB of host memory at 50 000 cells where int64 + fp64 host lists would take 40 GB
            Xs, corr = _search_space(self.fitdata, self.metric)
            idx, dist = ops.knn_search(Xs, sight, include_self=True)
            if corr:
                dist = dist * dist / 2.0
            self._lists_dev = (idx, dist)
            self.__dict__.pop("dsi", None); self.__dict__.pop("dist", None)
            logging.debug(f"Using the initialization network to find a {self.k}-NN graph with maximum connectivity of {self.maxl}")
            groups = None if self.constraint is None else np.asarray(self.constraint).astype("int64")
            self.dist_new, self.dsi_new, self.l = ops.balance_knn_device_lists(idx, dist, self.maxl, self.k, groups)
            if mode == "connectivity":
                self.dist = np.ones((idx.shape[0], idx.shape[1]), dtype=np.int64)
                self.dist[:, 0] = 0
            return self.dist_new, self.dsi_new, self.l
        logging.debug(f"Using the initialization network to find a {self.k}-NN graph with maximum connectivity of {self.maxl}")
        self.dist_new, self.dsi_new, self.l = knn_balance(self.dsi, self.dist, maxl=self.maxl, k=self.k, constraint=self.constraint)
        if mode == "connectivity":
            self.dist = np.ones_like(self.dsi)
            self.dist[:, 0] = 0
        return self.dist_new, self.dsi_new, self.l

    def kneighbors_graph(self, X: np.ndarray = None, maxl: int = None, mode: str = "distance") -> sparse.csr_matrix:
        """neighbors.py:291-322: CSR with k+1 stored entries per row (self first, distance 0 stored)."""
        dist_new, dsi_new, l = self.kneighbors(X=X, maxl=maxl, mode=mode)
        self.bknn = sparse.csr_matrix((np.ravel(dist_new), np.ravel(dsi_new),
                                       np.arange(0, dist_new.shape[0] * dist_new.shape[1] + 1, dist_new.shape[1])),
                                      (self.n_samples, self.n_samples))
        return self.bknn

    def smooth_data(self, data_to_smooth: np.ndarray, X: np.ndarray = None, maxl: int = None, mutual: bool = False,
                    only_increase: bool = True) -> np.ndarray:
        """neighbors.py:324-357."""
        if self.bknn is None:
            assert (X is None) and (maxl is None), "graph was already fit with different parameters"
            self.kneighbors_graph(X=X, maxl=maxl, mode=self.mode)
        connectivity = (self.bknn > 0).minimum((self.bknn > 0).T) if mutual else (self.bknn.T > 0)
        connectivity = connectivity.tolil()
        connectivity.setdiag(1)
        w = connectivity_to_weights(connectivity)          # rows sum to 1; the reference uses its transpose on the right
        if data_to_smooth.shape[1] == w.shape[0]:
            result = convolve_by_sparse_weights(data_to_smooth, w)
        elif data_to_smooth.shape[0] == w.shape[0]:
            result = convolve_by_sparse_weights(data_to_smooth.T, w).T
        else:
            raise ValueError(f"Incorrect size of matrix, none of the axis correspond to the one of graph. {w.shape}")
        return np.maximum(result, data_to_smooth) if only_increase else result


def connectivity_to_weights(mknn: sparse.spmatrix, axis: int = 1) -> sparse.csr_matrix:
    """neighbors.py:385-390: scale each row (axis=1) of a connectivity matrix to sum 1."""
    if not sparse.isspmatrix_csr(mknn):
        mknn = sparse.csr_matrix(mknn)
    return mknn.multiply(1.0 / sparse.csr_matrix.sum(mknn, axis=axis)).tocsr()


def _csr_parts(w: sparse.spmatrix):
    w = sparse.csr_matrix(w)
    return w.indptr.astype(np.int64), w.indices.astype(np.int32), np.ascontiguousarray(w.data, dtype=np.float64), w.shape


def convolve_by_sparse_weights(data, w: sparse.spmatrix, dtype=None, as_device: bool = False):
    """neighbors.py:416-423: ``data @ w.T`` for data (genes, cells) and a (cells, cells) weight matrix
    whose rows sum to one.  Returns a Fortran-ordered (genes, cells) fp64 array like the reference
    (or the device matrix when as_device=True)."""
    indptr, indices, vals, shape = _csr_parts(w)
    rowsum = np.asarray(sparse.csr_matrix(w).sum(1)).ravel()
    assert np.allclose(rowsum, 1), "weight matrix need to sum to one over the columns"
    D = CellMatrix.from_genes_major(data, dtype)
    if shape != (D.C, D.C):
        raise ValueError(f"weights {shape} do not match {D.C} cells")
    out = ops.knn_pool(D, indptr, indices, vals)
    return out if as_device else out.to_genes_major(order="F")


# --------------------------------------------------------------------------- mutual-kNN smoothing helpers (unused by VelocytoLoom)
def make_mutual(knn: sparse.spmatrix) -> sparse.spmatrix:
    """neighbors.py:379-382: element-wise minimum with the transpose, i.e. an edge survives only if both directions exist
    (distance graph: the smaller of the two distances)."""
    return sparse.csr_matrix(knn).minimum(sparse.csr_matrix(knn).T)


def min_n(row_data: np.ndarray, row_indices: np.ndarray, n: int) -> Tuple[np.ndarray, np.ndarray]:
    """neighbors.py:393-400: the n smallest stored entries of one sparse row and their column indices."""
    i = np.argsort(row_data)[:n]
    return row_data[i], row_indices[i]


def take_top(matrix: sparse.spmatrix, n: int) -> sparse.lil_matrix:
    """neighbors.py:403-411: keep the n smallest stored entries of every row (in ascending order of value)."""
    m = sparse.csr_matrix(matrix)
    out = sparse.lil_matrix(m.shape, dtype=m.dtype)
    for r in range(m.shape[0]):
        lo, hi = m.indptr[r], m.indptr[r + 1]
        d, c = min_n(m.data[lo:hi], m.indices[lo:hi], n)
        out.data[r], out.rows[r] = d.tolist(), c.tolist()
    return out


def knn_smooth_weights(matrix: np.ndarray, metric: str = "euclidean", k_search: int = 20, k_mutual: int = 10, n_jobs: int = 10
                       ) -> Tuple[sparse.spmatrix, sparse.csr_matrix]:
    """neighbors.py:426-451: smoothing weights from the mutual part of a kNN graph ((genes, cells) input; the search is the
    HIP kernel).  Returns (weights, knn)."""
    assert k_search >= k_mutual, "k_search needs to be bigger than k_mutual"
    knn = knn_distance_matrix(np.asarray(matrix).T, metric=metric, k=k_search, mode="distance", n_jobs=n_jobs)
    top_mknn = take_top(make_mutual(knn), k_mutual)
    top_mknn.setdiag(1)
    return connectivity_to_weights(top_mknn > 0), knn
