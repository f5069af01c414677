"""Drop-in for the numeric methods of ``velocyto.analysis.VelocytoLoom`` (analysis.py:26-2342).

Same method names, keyword arguments, defaults and attribute names as the reference; every
(genes x cells) matrix attribute lives on the MI355X as a cells-major ``ops.CellMatrix`` and is
converted to the reference's numpy ``(genes, cells)`` float64 array only when it is read
(``vlm.Sx``) - assigning a numpy array to such an attribute uploads it.  All hot loops are the HIP
kernels of libvelocyto_hip.so; nothing falls back to the CPU.

Covered: normalize/_normalize_*, knn_imputation[_precomputed], fit_gammas (all
weight modes and fit branches), predict_U, calculate_velocity, calculate_shift,
extrapolate_cell_at_t, estimate_transition_prob (knn_random and full, four transforms, randomised
control), calculate_embedding_shift, prepare_markov, run_markov, plus the "next" rows calculate_grid_arrows,
filter_genes_by_phase_portrait / filter_genes_good_fit and HDF5 (de)serialisation.  The callers upstream of the path
(cell / gene filters, feature scores, size normalisations, PCA, the default_* drivers) live in ``preprocess.PreprocessMixin``.
Plotting is out of scope.

Known, deliberate differences from the reference (each has a test):
  * matrices are accepted in any memory order (the reference's F-order trap is gone);
  * ``corrcoef`` / ``transition_prob`` are kept compact (cells x neighbours) on the device and
    densified to (cells, cells) only when the attribute is read;
  * default weighted fit = exact box-constrained least squares instead of L-BFGS-B's stopping point;
  * ``permute_rows_nsign`` (randomised control) draws from torch's device RNG: statistical parity only,
    as with the reference's numba RNG.
"""
from __future__ import annotations

import logging
import warnings
from typing import Any, Dict, List, Optional, Tuple, Union

import numpy as np
import torch
from scipy import sparse

from . import ops
from .diffusion import Diffusion
from .neighbors import BalancedKNN, connectivity_to_weights, knn_distance_matrix
from .ops import CellMatrix
from .preprocess import PreprocessMixin, colormap_fun  # noqa: F401  (colormap_fun is module-level in the reference too)

_MATRIX_ATTRS = frozenset(["S", "U", "A", "S_sz", "U_sz", "S_norm", "U_norm", "Sx", "Ux", "Sx_sz", "Ux_sz", "Sx_norm", "Ux_norm",
                           "Upred", "velocity", "delta_S", "delta_S_rndm", "Sx_sz_t", "Sx_t"])
_LAZY_DENSE = frozenset(["corrcoef", "corrcoef_random", "transition_prob", "transition_prob_random", "tr", "embedding_knn"])
# scipy containers of the pooling graph: assembled from the device rows the first time somebody reads them (knn_imputation itself
# pools from the device rows and needs neither)
_LAZY_GRAPH = frozenset(["knn", "knn_smoothing_w"])


class VelocytoLoom(PreprocessMixin):
    """Device-resident counterpart of velocyto.VelocytoLoom (analysis.py:26-94).

    ``VelocytoLoom(loom_filepath)`` reads a .loom file (HDF5 via loom_io: h5py if present, else ctypes-bound libhdf5);
    ``VelocytoLoom.from_arrays(S, U, A=None, ca=None, ra=None)`` starts from in-memory layers."""

    def __new__(cls, *args, **kwargs):
        # the device-side containers exist from creation, not from __init__: serialization.load_hdf5 makes its object with
        # obj_class.__new__(obj_class) like the reference (serialization.py:107), whatever __init__ a subclass defines
        self = super().__new__(cls)
        object.__setattr__(self, "_dev", {})
        object.__setattr__(self, "_host", {})
        object.__setattr__(self, "_dtype", ops.resolve_dtype(None))
        return self

    def __init__(self, loom_filepath: str = None, dtype=None) -> None:
        object.__setattr__(self, "_dtype", ops.resolve_dtype(dtype))
        if loom_filepath is not None:
            from .loom_io import read_loom
            self.loom_filepath = loom_filepath
            layers, ca, ra = read_loom(loom_filepath)                 # analysis.py:56-64
            self._init_layers(layers["spliced"], layers["unspliced"], layers.get("ambiguous"), ca, ra)

    @classmethod
    def from_arrays(cls, S, U, A=None, ca: Dict = None, ra: Dict = None, dtype=None) -> "VelocytoLoom":
        self = cls(None, dtype=dtype)
        self._init_layers(S, U, A, ca, ra)
        return self

    def _init_layers(self, S, U, A, ca, ra) -> None:
        self.S = S
        self.U = U
        if A is not None:
            self.A = A
        self.ca = dict(ca) if ca is not None else {"CellID": np.arange(self.dev("S").C)}
        self.ra = dict(ra) if ra is not None else {"Gene": np.arange(self.dev("S").G)}
        self.initial_cell_size = ops.row_sums(self.dev("S")).cpu().numpy()      # analysis.py:66-67
        self.initial_Ucell_size = ops.row_sums(self.dev("U")).cpu().numpy()

    # ------------------------------------------------------------------ attribute plumbing
    def dev(self, name: str) -> CellMatrix:
        """The device matrix behind a (genes x cells) attribute."""
        try:
            return self._dev[name]
        except KeyError:
            raise AttributeError(f"{name} has not been computed yet") from None

    def __getattr__(self, name: str):
        if name in _MATRIX_ATTRS:
            d = object.__getattribute__(self, "_dev")
            if name in d:
                h = object.__getattribute__(self, "_host")
                if name not in h:
                    a = d[name].to_genes_major(order="F")            # same values/strides the reference ends up with
                    a.setflags(write=False)                          # a host COPY of device data: an in-place edit would be lost, so
                    h[name] = a                                      # it fails loudly - assign the whole attribute to change it
                return h[name]
        elif name in _LAZY_DENSE:
            return self._densify(name)
        elif name in _LAZY_GRAPH:
            lazy = object.__getattribute__(self, "__dict__").get("_graph_lazy")
            if lazy is not None:
                return self._materialise_graph(name, lazy)
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{name}'")

    def __setattr__(self, name: str, value) -> None:
        if name in _MATRIX_ATTRS:
            self._host.pop(name, None)
            count_layer = value if isinstance(value, ops.CountMatrix) and name in ("S", "U") else None    # device uint16 layer as is
            if count_layer is not None:
                value = count_layer.to_float(self._dtype)
            self._dev[name] = value if isinstance(value, CellMatrix) else CellMatrix.from_genes_major(np.asarray(value), self._dtype)
            st = self.__dict__
            if name in ("S", "U"):
                # loom layers are uint16 molecule counts: keep them as such on the device too, so that pooling can
                # gather 2-byte elements (S_sz = norm_factor * S is applied on the fly, ops.knn_pool_counts)
                counts = st.setdefault("_counts", {})
                counts.pop(name, None)
                st.setdefault("_sz_scale", {}).pop(name + "_sz", None)
                if count_layer is not None:
                    counts[name] = count_layer
                elif isinstance(value, np.ndarray) and ops.CountMatrix.representable(value):
                    counts[name] = ops.CountMatrix.from_genes_major(value)
            elif name in ("S_sz", "U_sz"):
                st.setdefault("_sz_scale", {}).pop(name, None)       # assigned by hand: no longer factor * counts
        else:
            object.__setattr__(self, name, value)

    def __delattr__(self, name: str) -> None:
        if name in _MATRIX_ATTRS:
            self._dev.pop(name)
            self._host.pop(name, None)
        else:
            object.__delattr__(self, name)

    def _set_dev(self, name: str, m: CellMatrix) -> None:
        self._host.pop(name, None)
        self._dev[name] = m

    def _materialise_graph(self, name: str, lazy: dict):
        """`knn` / `knn_smoothing_w` as the scipy matrices the reference stores (analysis.py:1005-1010), built from the device rows
        knn_imputation kept (lists sorted by cell number - the state (self.knn > 0) leaves the matrix in -, distances all positive):
        once, on first access; afterwards they are plain attributes."""
        from .neighbors import weights_from_sorted_knn
        if "host" not in lazy:
            lazy["host"] = (lazy["idx_s"].cpu().numpy(), lazy["dist_s"].cpu().numpy())
        idx_s, dist_s = lazy["host"]
        n, k = idx_s.shape
        if name == "knn":
            m = sparse.csr_matrix((dist_s.ravel(), idx_s.ravel(), np.arange(0, n * k + 1, k)), shape=(n, n))
            m.has_sorted_indices = True
        else:
            m = weights_from_sorted_knn(idx_s, lazy["diag"])
        self.__dict__[name] = m
        return m

    def _densify(self, name: str):
        """(cells, cells) numpy views of the compact device results (read-only convenience)."""
        st = self.__dict__
        if name in ("corrcoef", "corrcoef_random"):
            key = "_corr" + ("_random" if name.endswith("random") else "")
            if key not in st:
                raise AttributeError(name)
            v = st[key]
            if v.shape[1] == v.shape[0] and st.get("corr_calc") == "full":
                return v.double().cpu().numpy()
            return ops.scatter_rows(v, st["_neigh"], v.shape[0]).double().cpu().numpy()
        if name in ("transition_prob", "transition_prob_random"):
            key = "_tp" + ("_random" if name.endswith("random") else "")
            if key not in st:
                raise AttributeError(name)
            return ops.scatter_rows(st[key], st["_tp_ixs"], st[key].shape[0]).double().cpu().numpy()
        if name == "embedding_knn":
            # estimate_transition_prob's neighbour graph (analysis.py:1551, 1565-1568) as the reference exposes it; the kernels
            # use the compact device list, so the 0/1 CSR is only assembled when somebody reads it
            if "_neigh" not in st:
                raise AttributeError(name)
            ix = st["_neigh"].cpu().numpy().astype(np.int64)
            if st.get("corr_calc") == "full":
                ix = np.sort(ix, axis=1)                                          # sklearn's connectivity graph: canonical CSR
            C, k1 = ix.shape
            return sparse.csr_matrix((np.ones(C * k1), ix.ravel(), np.arange(0, C * k1 + 1, k1)), shape=(C, C))
        if name == "tr":
            if "_tr_dev" not in st:
                raise AttributeError(name)
            return sparse.csr_matrix(st["_tr_dev"].dense(torch.float64).cpu().numpy())
        raise AttributeError(name)

    # ------------------------------------------------------------------ a1 normalisation
    def _size_factor(self, M: CellMatrix, relative_size, target_size) -> Tuple[torch.Tensor, torch.Tensor, float]:
        cell_size = ops.row_sums(M) if relative_size is None else torch.as_tensor(np.asarray(relative_size, dtype=np.float64), device=M.t.device)
        avg = float(cell_size.mean()) if target_size is None else float(target_size)
        return cell_size, avg / cell_size, avg

    def _normalize_S(self, size: bool = True, log: bool = True, pcount: float = 1, relative_size: Any = None, target_size: Any = None) -> None:
        """analysis.py:535-551."""
        S = self.dev("S")
        if size:
            rel = relative_size if type(relative_size) is np.ndarray else None
            cs, fac, avg = self._size_factor(S, rel, target_size)
            self.cell_size, self.avg_size, self.norm_factor = cs.cpu().numpy(), avg, fac.cpu().numpy()
        else:
            fac, self.norm_factor = None, 1
        sz, nm = ops.scale_log(S, fac, True, log, pcount)
        self._set_dev("S_sz", sz)
        self.__dict__.setdefault("_sz_scale", {})["S_sz"] = fac if fac is not None else torch.ones(S.C, dtype=torch.float64, device=S.t.device)
        if log:
            self._set_dev("S_norm", nm)

    def _normalize_U(self, size: bool = True, log: bool = True, pcount: float = 1, use_S_size: bool = False,
                     relative_size: np.ndarray = None, target_size: Any = None) -> None:
        """analysis.py:553-582."""
        U = self.dev("U")
        if size:
            if use_S_size:
                rel = self.cell_size if hasattr(self, "cell_size") else ops.row_sums(self.dev("S")).cpu().numpy()
            elif type(relative_size) is np.ndarray:
                rel = relative_size
            else:
                rel = None
            cs, fac, avg = self._size_factor(U, rel, target_size)
            self.Ucell_size, self.Uavg_size, self.Unorm_factor = cs.cpu().numpy(), avg, fac.cpu().numpy()
        else:
            fac, self.Unorm_factor = None, 1
        sz, nm = ops.scale_log(U, fac, True, log, pcount, fix_nonfinite=True)
        self._set_dev("U_sz", sz)
        scale = fac if fac is not None else torch.ones(U.C, dtype=torch.float64, device=U.t.device)
        if bool(torch.isfinite(scale).all()):                         # (cells without unspliced counts: float path keeps the :580 fix-up)
            self.__dict__.setdefault("_sz_scale", {})["U_sz"] = scale
        else:
            self.__dict__.setdefault("_sz_scale", {}).pop("U_sz", None)
        if log:
            self._set_dev("U_norm", nm)

    def _normalize_Sx(self, size: bool = True, log: bool = True, pcount: float = 1, relative_size: Any = None, target_size: Any = None) -> None:
        """analysis.py:584-600."""
        Sx = self.dev("Sx")
        if size:
            cs, fac, avg = self._size_factor(Sx, relative_size if relative_size is not None and np.size(relative_size) > 0 and np.any(relative_size) else None, target_size)
            self.xcell_size, self.xavg_size, self.xnorm_factor = cs.cpu().numpy(), avg, fac.cpu().numpy()
        else:
            fac, self.xnorm_factor = None, 1
        sz, nm = ops.scale_log(Sx, fac, True, log, pcount)
        self._set_dev("Sx_sz", sz)
        if log:
            self._set_dev("Sx_norm", nm)

    def _normalize_Ux(self, size: bool = True, log: bool = True, pcount: float = 1, use_Sx_size: bool = False, relative_size: Any = None,
                      target_size: Any = None) -> None:
        """analysis.py:602-631."""
        Ux = self.dev("Ux")
        if size:
            if use_Sx_size:
                rel = self.xcell_size if hasattr(self, "cell_size") else ops.row_sums(self.dev("Sx")).cpu().numpy()
            elif type(relative_size) is np.ndarray:
                rel = relative_size
            else:
                rel = None
            cs, fac, avg = self._size_factor(Ux, rel, target_size)
            self.xUcell_size, self.xUavg_size, self.xUnorm_factor = cs.cpu().numpy(), avg, fac.cpu().numpy()
        else:
            fac, self.xUnorm_factor = None, 1
        sz, nm = ops.scale_log(Ux, fac, True, log, pcount, fix_nonfinite=True)
        self._set_dev("Ux_sz", sz)
        if log:
            self._set_dev("Ux_norm", nm)

    def normalize(self, which: str = "both", size: bool = True, log: bool = True, pcount: float = 1, relative_size: np.ndarray = None,
                  use_S_size_for_U: bool = False, target_size: Tuple[float, float] = (None, None)) -> None:
        """analysis.py:633-676."""
        if which == "both":
            self._normalize_S(size=size, log=log, pcount=pcount, relative_size=relative_size, target_size=target_size[0])
            self._normalize_U(size=size, log=log, pcount=pcount, use_S_size=use_S_size_for_U, relative_size=relative_size, target_size=target_size[1])
        if "S" == which:
            self._normalize_S(size=size, log=log, pcount=pcount, relative_size=relative_size, target_size=target_size[0])
        if "U" == which:
            self._normalize_U(size=size, log=log, pcount=pcount, use_S_size=use_S_size_for_U, relative_size=relative_size, target_size=target_size[1])
        if which == "imputed":
            self._normalize_Sx(size=size, log=log, pcount=pcount, relative_size=relative_size, target_size=target_size[0])
            self._normalize_Ux(size=size, log=log, pcount=pcount, use_Sx_size=use_S_size_for_U, relative_size=relative_size, target_size=target_size[1])
        if "Sx" == which:
            self._normalize_Sx(size=size, log=log, pcount=pcount, relative_size=relative_size, target_size=target_size[0])
        if "Ux" == which:
            self._normalize_Ux(size=size, log=log, pcount=pcount, use_Sx_size=use_S_size_for_U, relative_size=relative_size, target_size=target_size[1])

    # ------------------------------------------------------------------ stage A
    def knn_imputation(self, k: int = None, pca_space: float = True, metric: str = "euclidean", diag: float = 1, n_pca_dims: int = None,
                       maximum: bool = False, size_norm: bool = True, balanced: bool = False, b_sight: int = None, b_maxl: int = None,
                       group_constraint: Union[str, np.ndarray] = None, n_jobs: int = 8) -> None:
        """analysis.py:933-1023."""
        N = self.dev("S").C
        if k is None:
            k = int(N * 0.025)
        if b_sight is None and balanced:
            b_sight = np.maximum(int(k * 8), N - 1)
        if b_maxl is None and balanced:
            b_maxl = np.maximum(int(k * 4), N - 1)
        space = self.pcs[:, :n_pca_dims] if pca_space else self.S_norm.T
        w_direct = False                                          # knn_smoothing_w written out directly (below) instead of through scipy
        dev_rows = None                                           # ... and kept as device rows (unbalanced graph without duplicate cells)
        for stale in ("_graph_lazy", "knn", "knn_smoothing_w"):   # a previous graph, lazy or materialised, is gone
            self.__dict__.pop(stale, None)
        if balanced:
            constraint = None
            if group_constraint is not None:
                constraint = np.array(self.cluster_ix) if isinstance(group_constraint, str) and group_constraint == "clusters" else np.asarray(group_constraint)
            bknn = BalancedKNN(k=k, sight_k=b_sight, maxl=b_maxl, metric=metric, constraint=constraint, mode="distance", n_jobs=n_jobs)
            bknn.fit(space)
            self.knn = bknn.kneighbors_graph(mode="distance")
            if diag != 0:
                # every row = the cell itself (distance 0, first) + k distinct neighbours at positive distances - no padding, no
                # coincident cells: the graph column-sorted, as (self.knn > 0) leaves it, and the weights written out directly
                from .neighbors import weights_from_sorted_knn
                dsi, dst = np.asarray(bknn.dsi_new), np.asarray(bknn.dist_new)
                n_rows = dsi.shape[0]
                if dsi.shape[1] == k + 1 and np.array_equal(dsi[:, 0], np.arange(n_rows)) and bool((dst[:, 1:] > 0).all()):
                    order = np.argsort(dsi, axis=1, kind="stable")
                    cols = np.take_along_axis(dsi, order, 1)
                    self.knn = sparse.csr_matrix((np.take_along_axis(dst, order, 1).ravel(), cols.ravel().astype(np.int32),
                                                  np.arange(0, n_rows * (k + 1) + 1, k + 1)), shape=(n_rows, n_rows))
                    self.knn.has_sorted_indices = True
                    others = cols[cols != np.arange(n_rows)[:, None]].reshape(n_rows, k)       # the row without the cell itself, still sorted
                    self.knn_smoothing_w = weights_from_sorted_knn(others, diag)
                    w_direct = True
        else:
            if group_constraint is not None:
                raise ValueError("group_constraint is currently supported only if the argument balanced is set to True")
            if diag != 0:
                # the graph with its rows sorted by cell number on the device - the state (self.knn > 0) leaves self.knn in, :1006 -
                # and, when no distance is zero (no duplicate cells), the weights written out directly ON THE DEVICE: the same
                # matrices as the scipy chain below (tests/test_host_and_abi.py, tests/test_gpu_facade.py), none of which is built
                # unless somebody reads `vlm.knn` / `vlm.knn_smoothing_w` (the containers cost 25 ms of host time at 50 000 cells
                # against 12 ms of kernels)
                from .neighbors import _kneighbors_rows_device
                idx_s, dist_s, positive = _kneighbors_rows_device(space, k, metric)
                if positive:
                    n_rows = int(idx_s.shape[0])
                    self.__dict__["_graph_lazy"] = {"idx_s": idx_s, "dist_s": dist_s, "diag": diag}
                    dev_rows = ops.weight_rows_from_sorted_knn(idx_s, diag, self._dtype)
                    w_direct = True
                else:
                    n_rows = int(idx_s.shape[0])
                    self.knn = sparse.csr_matrix((dist_s.cpu().numpy().ravel(), idx_s.cpu().numpy().ravel(), np.arange(0, n_rows * k + 1, k)),
                                                 shape=(n_rows, n_rows))
                    self.knn.has_sorted_indices = True
            else:
                self.knn = knn_distance_matrix(space, metric=metric, k=k, mode="distance", n_jobs=n_jobs)
        if not w_direct:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                connectivity = (self.knn > 0).astype(float)      # :1006 (also column-sorts self.knn in place, like scipy does there)
                connectivity.setdiag(diag)
            self.knn_smoothing_w = connectivity_to_weights(connectivity)
        # schedule the pooling along the Hilbert curve of the two leading coordinates of the search space (results do not
        # depend on it; neighbouring cells gather overlapping rows while they are still in L2)
        self.__dict__["_pool_order"] = ops.hilbert_order(np.ascontiguousarray(space[:, :2])) if np.shape(space)[1] >= 2 else None
        self._pool(self.knn_smoothing_w if dev_rows is None else dev_rows, maximum and size_norm, "S_sz" if size_norm else "S", "U_sz" if size_norm else "U")
        if maximum and not size_norm:
            # the reference takes the maximum with the SIZE-NORMALISED layers whatever was pooled (analysis.py:1017-1019:
            # np.maximum(self.S_sz, self.Sx)); reproduced as is
            for pooled, raw in (("Sx", "S_sz"), ("Ux", "U_sz")):
                m = CellMatrix(torch.maximum(self.dev(pooled).t, self.dev(raw).t), self.dev(pooled).G)
                self._set_dev(pooled, m)
                self._set_dev(pooled + "_sz", m)

    def knn_imputation_precomputed(self, knn_smoothing_w: sparse.spmatrix, maximum: bool = False) -> None:
        """analysis.py:1025-1053."""
        self.__dict__["_pool_order"] = None
        self._pool(knn_smoothing_w, maximum, "S_sz", "U_sz")

    def _pool(self, w, maximum: bool, s_name: str, u_name: str) -> None:
        """w: the weight matrix (scipy sparse, rows summing to one) or its rows already on the device as (indptr, indices, values)."""
        if isinstance(w, tuple):
            indptr, indices, vals = w
            n_rows = int(indptr.numel()) - 1
        else:
            w = sparse.csr_matrix(w)
            assert np.allclose(np.asarray(w.sum(1)).ravel(), 1), "weight matrix need to sum to one over the columns"   # neighbors.py:422
            indptr, indices, vals = w.indptr.astype(np.int64), w.indices.astype(np.int32), np.ascontiguousarray(w.data, dtype=np.float64)
            n_rows = w.shape[0]
        counts, scale = self.__dict__.get("_counts", {}), self.__dict__.get("_sz_scale", {})
        order = self.__dict__.get("_pool_order")
        if order is not None and int(order.numel()) != n_rows:
            order = None
        if "S" in counts and "U" in counts and ((s_name, u_name) == ("S", "U") or
                                                ((s_name, u_name) == ("S_sz", "U_sz") and "S_sz" in scale and "U_sz" in scale)):
            sS, sU = (None, None) if s_name == "S" else (scale["S_sz"], scale["U_sz"])
            if counts["S"].t.dtype != counts["U"].t.dtype:      # one uint8, one uint16 layer: keep both as uint16 from now on
                for n in ("S", "U"):
                    if counts[n].t.dtype == torch.uint8:
                        counts[n] = ops.CountMatrix(counts[n].t.to(torch.int16), counts[n].G)
            Sx, Ux = ops.knn_pool_counts(counts["S"], counts["U"], sS, sU, indptr, indices, vals, dtype=self._dtype, maximum=maximum, order=order,
                                         validate=not isinstance(w, tuple))
        else:
            Sx, Ux = ops.knn_pool2(self.dev(s_name), self.dev(u_name), indptr, indices, vals, maximum=maximum, order=order)
        self._set_dev("Sx", Sx)
        self._set_dev("Ux", Ux)
        # :1022-1023 makes separate copies "for backwards compatibility".  Device matrices are never edited in place through the
        # attributes (hosts views are read-only copies, assignment replaces the entry), so the two names share one matrix until one
        # of them is rescaled (preprocess._scale_cells copies first): 12 GB less to allocate and to write at 50 000 x 30 000
        self._set_dev("Sx_sz", Sx)
        self._set_dev("Ux_sz", Ux)

    # ------------------------------------------------------------------ stage B
    def fit_gammas(self, steady_state_bool: np.ndarray = None, use_imputed_data: bool = True, use_size_norm: bool = True,
                   fit_offset: bool = True, fixperc_q: bool = False, weighted: bool = True, weights: Any = "maxmin_diag",
                   limit_gamma: bool = False, maxmin_perc: List[float] = [2, 98], maxmin_weighted_pow: float = 15) -> None:
        """analysis.py:1120-1260."""
        from .estimation import _fixperc_q, _median_and_up_gamma, _weighted_offset_device
        C_all = self.dev("S").C
        # analysis.py:1159-1162 stores the mask and :1223-1257 fits on tmpS[:, steady_state], tmpU[:, steady_state].  Two things of
        # the reference are fixed here rather than reproduced: `if steady_state_bool:` raises on an array of
        # more than one element (a list works), and the weights W - built from percentiles over ALL cells, :1179-1219 - are not
        # subset, so the weighted fits fail on broadcasting as soon as the mask drops a cell.  Here a boolean mask of length
        # `cells` or an array of cell indices is taken (both work as `tmpS[:, steady_state]` in the reference), the thresholds / weights
        # are computed over all cells as in the reference and restricted to the same cells.
        ss_rows = None
        if steady_state_bool is None or np.ndim(steady_state_bool) == 0:
            # a scalar: the reference stores it and indexes with it (True adds an axis and the fits fail); taken as "all cells"
            self.steady_state = np.ones(C_all, dtype=bool)
        else:
            ss = np.asarray(steady_state_bool)
            if ss.dtype == np.bool_:
                if ss.shape != (C_all,):
                    raise ValueError(f"steady_state_bool must be a boolean mask over the {C_all} cells (or an array of cell indices)")
                if not ss.any():
                    raise ValueError("steady_state_bool selects no cell")
                rows = None if ss.all() else np.flatnonzero(ss)
            elif np.issubdtype(ss.dtype, np.integer) and ss.ndim == 1:
                # tmpS[:, self.steady_state] with an index array (analysis.py:1223-1257): numpy's fancy indexing - any order, repeats
                # count twice, negative numbers from the end
                if ss.size == 0:
                    raise ValueError("steady_state_bool selects no cell")
                if ss.min() < -C_all or ss.max() >= C_all:
                    raise IndexError(f"steady_state_bool: cell index out of range for {C_all} cells")
                rows = np.where(ss < 0, ss + C_all, ss).astype(np.int64)
            else:
                raise ValueError(f"steady_state_bool must be a boolean mask over the {C_all} cells or a 1-d array of cell indices")
            self.steady_state = ss
            if rows is not None:
                ss_rows = torch.from_numpy(rows).to(self.dev("S").t.device)
        if use_imputed_data:
            tmpS, tmpU = (self.dev("Sx_sz"), self.dev("Ux_sz")) if use_size_norm else (self.dev("Sx"), self.dev("Ux"))
        else:
            tmpS, tmpU = (self.dev("S_sz"), self.dev("U_sz")) if use_size_norm else (self.dev("S"), self.dev("U"))
        perc = [float(p) for p in maxmin_perc]
        wmode, wargs = 2, {}
        if weighted:
            if type(weights) is np.ndarray:
                wmode, wargs = 0, dict(W=CellMatrix.from_genes_major(weights, tmpS.dtype))
            elif weights in ("sum", "prod"):
                pS, pU = ops.gene_quantiles(tmpS, [99])[0], ops.gene_quantiles(tmpU, [99])[0]
                wmode, wargs = 0, dict(W=ops.gamma_weights(tmpS, tmpU, 0 if weights == "sum" else 1, pS, pU))
            elif weights == "maxmin_weighted":
                q = ops.gene_quantiles(tmpS, perc)
                wmode, wargs = 0, dict(W=ops.gamma_weights(tmpS, None, 2, q[0], q[1], power=maxmin_weighted_pow))
            elif weights == "maxmin":
                q = ops.gene_quantiles(tmpS, perc)
                wmode, wargs = 1, dict(M=tmpS, down=q[0], up=q[1])
            elif weights in ("maxmin_diag", "maxmin_double"):
                Sx, Ux = self.dev("Sx"), self.dev("Ux")
                dS, dU = self._maxnorm_denominator(Sx), self._maxnorm_denominator(Ux)
                q = ops.gene_quantiles(Sx, perc, M2=Ux, scale_a=dS, scale_b=dU)
                if weights == "maxmin_diag":
                    wmode, wargs = 1, dict(M=Sx, M2=Ux, scale_a=dS, scale_b=dU, down=q[0], up=q[1])
                else:
                    q2 = ops.gene_quantiles(Sx, perc)
                    wmode, wargs = 0, dict(W=ops.gamma_weights(Sx, Ux, 3, q[0], q[1], q2[0], q2[1], dS, dU))
            else:
                raise ValueError(f"weights={weights!r} is not supported")
        if ss_rows is not None:                                   # cells-major: the subset of cells is a gather of rows
            take = lambda m: CellMatrix(m.t.index_select(0, ss_rows).contiguous(), m.G)
            cache = {}
            def sub(m):
                if id(m) not in cache:
                    cache[id(m)] = take(m)
                return cache[id(m)]
            tmpS, tmpU = sub(tmpS), sub(tmpU)
            wargs = {k: (sub(v) if isinstance(v, CellMatrix) else v) for k, v in wargs.items()}
        R2 = None
        if fit_offset:
            if weighted:
                g, q, R2 = _weighted_offset_device(tmpU, tmpS, wmode, wargs, False, limit_gamma)
            else:
                if limit_gamma:
                    logging.warning("limit_gamma not implemented with this settings")
                g, q, _ = ops.fit_weighted(tmpU, tmpS, 2, fit_offset=True, box_q=False, want_R2=False)
        elif fixperc_q:
            if weighted:
                g, q, _ = _weighted_offset_device(tmpU, tmpS, wmode, wargs, True, limit_gamma)
            else:
                if limit_gamma:
                    logging.warning("limit_gamma not implemented with this settings")
                g, q, _ = ops.fit_weighted(tmpU, tmpS, 2, fit_offset=False, lo_gamma=0.0, up_gamma_default=20.0, q_fixed=_fixperc_q(tmpU, tmpS), want_R2=False)
        else:
            if weighted:
                if limit_gamma:
                    g, _, R2 = ops.fit_weighted(tmpU, tmpS, wmode, fit_offset=False, lo_gamma=1e-8, up_gamma=_median_and_up_gamma(tmpU, tmpS), **wargs)
                else:
                    g, _, R2 = ops.fit_weighted(tmpU, tmpS, wmode, fit_offset=False, lo_gamma=0.0, up_gamma_default=20.0, **wargs)
            else:
                if limit_gamma:
                    logging.warning("limit_gamma not implemented with this settings")
                g = ops.fit_slope(tmpU, tmpS)
            q = torch.zeros_like(g)
        g = torch.where(torch.isfinite(g), g, torch.zeros_like(g))                     # :1260
        self._gammas_dev, self._q_dev = g, q
        self.gammas, self.q = g.cpu().numpy(), q.cpu().numpy()
        if R2 is not None:
            self.R2 = R2.cpu().numpy()

    @staticmethod
    def _maxnorm_denominator(M: CellMatrix) -> torch.Tensor:
        """percentile(M, 99.9) with zeros replaced by max(max(row), 0.001)   (analysis.py:1197-1202)."""
        q = ops.gene_quantiles(M, [99.9, 100])
        return torch.where(q[0] == 0, torch.clamp(q[1], min=0.001), q[0])

    # ------------------------------------------------------------------ stage C
    def _chain(self, want: Tuple[str, ...]) -> Dict[str, CellMatrix]:
        st = self.__dict__
        which_S = st.get("which_S_for_pred", "Sx_sz")
        Sd = self.dev(which_S)
        Ud = self.dev("Ux_sz" if which_S == "Sx_sz" else "Ux")
        gam = torch.as_tensor(np.asarray(getattr(self, st.get("_which_gamma", "gammas")), dtype=np.float32))
        off = st.get("_which_offset", "q")
        q = None if off is None else torch.as_tensor(np.asarray(getattr(self, off), dtype=np.float32))
        eps_thr = None
        if st.get("_vel_eps"):
            up = ops.velocity_chain(Sd, Ud, gam, q, want=("Upred",))["Upred"]
            eps_thr = ops.gene_quantiles(up, [100])[0] * float(st["_vel_eps"])   # Upred.max(1) * eps   (:1378)
        return ops.velocity_chain(Sd, Ud, gam, q, want=want, eps_thr=eps_thr, dt_shift=st.get("_shift_dt", 1.0),
                                  dt_extrap=st.get("_extrap_dt", 1.0), used_dt=st.get("used_delta_t", 1.0),
                                  assumption=st.get("_assumption", 0), clip=st.get("_clip", True))

    def predict_U(self, which_gamma: str = "gammas", which_S: str = "Sx_sz", which_offset: str = "q") -> None:
        """analysis.py:1321-1346."""
        self.which_S_for_pred = which_S
        self._which_gamma, self._which_offset = which_gamma, which_offset
        if which_offset is None and (hasattr(self, "q_W") or hasattr(self, "q")):
            logging.warning("Predicting U without intercept but intercept was previously fit! Set which_offset='q' or 'q_W' ")
        self._set_dev("Upred", self._chain(("Upred",))["Upred"])

    def calculate_velocity(self, kind: str = "residual", eps: float = None) -> None:
        """analysis.py:1348-1379."""
        if kind != "residual":
            raise NotImplementedError(f"Velocity calculation kind={kind} is not implemented")
        if self.which_S_for_pred not in ("Sx_sz", "Sx"):
            raise NotImplementedError(f"Not implemented with which_S = {self.which_S_for_pred}")
        self._vel_eps = eps
        # velocity = Ux_sz - self.Upred with the STORED Upred (analysis.py:1369-1379): an Upred edited by the user propagates
        Ud, up = self.dev("Ux_sz" if self.which_S_for_pred == "Sx_sz" else "Ux"), self.dev("Upred")
        thr = ops.gene_quantiles(up, [100])[0] * float(eps) if eps else None           # Upred.max(1) * eps   (:1378)
        self._set_dev("velocity", ops.lincomb(Ud, up, 1.0, -1.0, zero_below=thr))

    def calculate_shift(self, assumption: str = "constant_velocity", delta_t: float = 1) -> None:
        """analysis.py:1381-1408."""
        if assumption not in ("constant_velocity", "constant_unspliced"):
            raise NotImplementedError(f"Assumption {assumption} is not implemented")
        self._assumption, self._shift_dt = (0 if assumption == "constant_velocity" else 1), float(delta_t)
        if assumption == "constant_velocity":
            self._set_dev("delta_S", ops.lincomb(self.dev("velocity"), None, float(delta_t)))     # delta_t * self.velocity (:1399)
        else:
            self._set_dev("delta_S", self._chain(("delta_S",))["delta_S"])          # closed form from Sx_sz, Ux_sz, gammas, q (:1400-1406)

    def extrapolate_cell_at_t(self, delta_t: float = 1, clip: bool = True) -> None:
        """analysis.py:1410-1439."""
        self._extrap_dt, self._clip = float(delta_t), bool(clip)
        if clip:
            self.used_delta_t = delta_t
        if self.which_S_for_pred not in ("Sx_sz", "Sx"):
            raise NotImplementedError("not implemented for other situations other than Sx or Sx_sz")
        # Sx_sz + delta_t * self.delta_S on the STORED delta_S, clipped at 0 (:1429-1431)
        out = ops.lincomb(self.dev(self.which_S_for_pred), self.dev("delta_S"), 1.0, float(delta_t), clip=clip)
        self._set_dev("Sx_sz_t" if self.which_S_for_pred == "Sx_sz" else "Sx_t", out)

    # ------------------------------------------------------------------ stage D
    def estimate_transition_prob(self, hidim: str = "Sx_sz", embed: str = "ts", transform: str = "sqrt", ndims: int = None,
                                 n_sight: int = None, psc: float = None, knn_random: bool = True, sampled_fraction: float = 0.3,
                                 sampling_probs: Tuple[float, float] = (0.5, 0.1), max_dist_embed: float = None, n_jobs: int = 4,
                                 threads: int = None, calculate_randomized: bool = True, random_seed: int = 15071990, **kwargs) -> None:
        """analysis.py:1452-1668."""
        self.which_hidim = hidim
        n_neighbors = kwargs.pop("n_neighbors", None)
        # extension (not in the reference): draw the neighbour subsample on the device instead of replaying numpy's
        # legacy RNG stream (0.3 s at 50k cells); same distribution, different random numbers
        device_sampling = bool(kwargs.pop("device_sampling", False))
        if kwargs:
            logging.warning(f"keyword arguments were passed but could not be interpreted {kwargs}")
        C = self.dev("S").C
        if n_sight is None and n_neighbors is None:
            n_neighbors = int(C / 5)
        if (n_sight is not None) and (n_neighbors is not None) and n_neighbors != n_sight:
            raise ValueError("n_sight and n_neighbors are different names for the same parameter, they cannot be set differently")
        if n_sight is not None and n_neighbors is None:
            n_neighbors = n_sight
        if psc is None:
            psc = 1.0 if transform in ("log", "logratio") else (1e-10 if transform == "sqrt" else 0)
        if transform not in ("log", "logratio", "linear", "sqrt"):
            raise NotImplementedError(f"transform={transform} is not a valid parameter")
        if "pcs" in hidim:
            raise NotImplementedError("hidim='pcs' (velocity in PCA space) is not on the accelerated path")
        if ndims is not None:
            raise ValueError(f"ndims was set to {ndims} but hidim != 'pcs'. Set ndims = None for hidim='{hidim}'")
        if knn_random:
            np.random.seed(random_seed)                                            # :1529
        hi = self.dev(hidim)
        dS = self.dev("delta_S")
        embedding = np.asarray(getattr(self, embed), dtype=np.float64)
        self.embedding = embedding
        mode = {"linear": 0, "sqrt": 1, "log": 2, "logratio": 3}[transform]
        kern = {"linear": ops.LINEAR, "sqrt": ops.SQRT, "log": ops.LOG10, "logratio": ops.LINEAR}[transform]
        # (uploaded before any kernel is queued: a host-to-device copy waits for what is ahead of it in the stream, and the host with it)
        emb_dev = torch.from_numpy(np.ascontiguousarray(embedding)).to(hi.t.device)
        # f32 sqrt with a negligible pseudocount on a matrix of ordinary scale: the three-instruction form (decided from
        # whole-matrix reductions); `vlm.literal_rule = True` (or VELOCYTO_AMD_LITERAL_RULE=1) keeps the literal rule.
        # (Decided before anything is queued - it reads a scalar back - so that the host does not wait for the device on its way to
        # the sampling; for logratio the matrix it looks at is only made below.)
        literal = bool(getattr(self, "literal_rule", False))
        rules = ops.partial_rules_for(hi, kern, psc, literal=literal) if knn_random and transform != "logratio" else None
        dmat_buf = dmat_r_buf = None
        self._dev.pop("delta_S_rndm", None)                       # a previous control goes back to the allocator before the new one is made
        self._host.pop("delta_S_rndm", None)
        if calculate_randomized:
            # the randomised control first (:1540-1541): its gene-major shuffle borrows the two buffers the transforms below fill
            dmat_buf, dmat_r_buf = (CellMatrix(torch.empty_like(hi.t), hi.G) for _ in range(2))
            self._set_dev("delta_S_rndm", _permute_rows_nsign(dS, random_seed, scratch=(dmat_buf.t, dmat_r_buf.t)))
        dmat, e_alt = ops.delta_transform(hi, dS, self.used_delta_t, mode, psc, out=dmat_buf)     # :1538, 1575-1601
        e = e_alt if transform == "logratio" else hi
        if knn_random and rules is None:
            rules = ops.partial_rules_for(e, kern, psc, literal=literal)
        dmat_r = None
        if calculate_randomized:
            dmat_r, _ = ops.delta_transform(hi, self.dev("delta_S_rndm"), self.used_delta_t, mode, psc, out=dmat_r_buf)
        # embedding kNN, n_neighbors + 1 nearest (query excluded)                    :1547-1549
        knn_ix, _ = ops.knn_search(emb_dev, n_neighbors + 1, include_self=False)
        if knn_random:
            self.corr_calc = "knn_random"
            n_cand = int(knn_ix.shape[1])
            p = np.linspace(sampling_probs[0], sampling_probs[1], n_cand)
            p = p / p.sum()
            size = int(sampled_fraction * (n_neighbors + 1))
            dev = hi.t.device
            self.__dict__.pop("embedding_knn", None)
            sched = ops.hilbert_order(emb_dev) if embedding.shape[1] >= 2 else None        # scheduling only: same numbers in any order
            self.__dict__["_embed_order"] = sched
            corr = torch.empty((C, size), dtype=hi.dtype, device=dev)
            corr_r = torch.empty((C, size), dtype=hi.dtype, device=dev) if calculate_randomized else None

            def correlate(nb, c0, c1, order, presorted=None):
                # the reference's two colDeltaCor*partial calls (:1578-1601) share e and the neighbour lists, hence every
                # A = f(e_i - e_c): one dual-control pass instead of two launches (vcy_coldeltacor_partial_dual)
                if size == 0 or c1 == c0:
                    return
                if calculate_randomized:
                    ops.coldeltacor_partial_dual(e, dmat, dmat_r, nb, kern, rules, psc, cell0=c0, order=order, out=corr[c0:c1],
                                                 out_rndm=corr_r[c0:c1], validate=False, presorted=presorted)
                else:
                    ops.coldeltacor_partial(e, dmat, nb, kern, rules, psc, cell0=c0, order=order, out=corr[c0:c1], validate=False,
                                            presorted=presorted)

            if device_sampling:
                # weighted sampling without replacement (Efraimidis-Spirakis keys u^(1/p), top `size`) with torch's device RNG
                gen = torch.Generator(device=dev).manual_seed(int(random_seed))
                keys = torch.log(torch.rand((C, n_cand), generator=gen, device=dev, dtype=torch.float64)) / \
                    torch.as_tensor(p, device=dev)[None, :]
                picks = torch.topk(keys, size, dim=1).indices
                sampling_ixs = picks.cpu().numpy()
                neigh = torch.gather(knn_ix, 1, picks).to(torch.int32).contiguous()       # neigh_ixs[arange(C)[:, None], sampling_ixs]  (:1565)
                correlate(neigh, 0, C, sched)
            else:
                # identical numpy legacy-RNG stream to the reference (:1561-1564), replayed on the host in blocks of cells; the
                # correlations of a block are launched as soon as its rows are drawn, so the device works through stage D while
                # the (sequential) replay goes on
                neigh = torch.empty((C, size), dtype=torch.int32, device=dev)
                rank = None
                if sched is not None:
                    rank = torch.empty(C, dtype=torch.int64, device=dev)
                    rank[sched.long()] = torch.arange(C, device=dev)

                # a block's rows go up on a side stream: on the current one the copy would queue behind the stage-D launch of the
                # previous block and hold the replay up until that launch is through
                copy_stream = torch.cuda.Stream(device=dev)

                def on_block(rows, c0, c1):
                    with torch.cuda.stream(copy_stream):
                        picks = torch.from_numpy(rows[c0:c1]).to(dev)
                    torch.cuda.current_stream().wait_stream(copy_stream)
                    picks.record_stream(torch.cuda.current_stream())
                    nb = torch.gather(knn_ix[c0:c1], 1, picks).to(torch.int32).contiguous()
                    neigh[c0:c1] = nb
                    # (presorted=False: sampled rows are in draw order; saying so spares the launch wrapper a look at the device)
                    correlate(nb, c0, c1, None if rank is None else torch.argsort(rank[c0:c1]).to(torch.int32), presorted=False)

                sampling_ixs = ops.choice_stream_host(n_cand, size, p, C, on_block=on_block)
            self.sampling_ixs = sampling_ixs
            self._neigh = neigh
            self._corr = corr
            if ops.corr_fixup(self._corr, neigh, zero_self=True, fix_nan=True, nan_to=1.0):                      # :1604-1607
                logging.warning("Nans encountered in corrcoef and corrected to 1s. If not identical cells were present it is probably a small isolated cluster converging after imputation.")
            if calculate_randomized:
                self._corr_random = corr_r
                if ops.corr_fixup(self._corr_random, neigh, zero_self=True, fix_nan=True, nan_to=1.0):
                    logging.warning("Nans encountered in corrcoef_random and corrected to 1s. If not identical cells were present it is probably a small isolated cluster converging after imputation.")
            else:
                self.__dict__.pop("_corr_random", None)
        else:
            self.corr_calc = "full"
            self.__dict__.pop("embedding_knn", None)
            self._neigh = knn_ix
            self._corr = ops.coldeltacor_full(e, dmat, kern, psc)
            _fill_diagonal_zero(self._corr)                                          # :1666 (off-diagonal NaNs are kept)
            if calculate_randomized:
                self._corr_random = ops.coldeltacor_full(e, dmat_r, kern, psc)
                _fill_diagonal_zero(self._corr_random)
            else:
                self.__dict__.pop("_corr_random", None)

    # ------------------------------------------------------------------ stage E
    def calculate_embedding_shift(self, sigma_corr: float = 0.05, expression_scaling: bool = True, scaling_penalty: float = 1.) -> None:
        """analysis.py:1670-1733, in neighbour-list form (no (cells, cells) temporaries)."""
        if self.corr_calc not in ("full", "knn_random"):
            raise NotImplementedError(f"Weird value self.corr_calc={self.corr_calc}")
        neigh = self._neigh
        hi = self.dev(self.which_hidim)
        dev = hi.t.device

        order = self.__dict__.get("_embed_order") if self.corr_calc == "knn_random" else None
        n = neigh.shape[1]
        names = [("_corr", "delta_S")] + ([("_corr_random", "delta_S_rndm")] if "_corr_random" in self.__dict__ else [])
        parts = []
        for corr_name, _ in names:
            corr = self.__dict__[corr_name]
            c = corr if self.corr_calc == "knn_random" else torch.gather(corr, 1, neigh.long())
            parts.append(ops.transition_prob(c.contiguous(), neigh, self.embedding, sigma_corr))       # (tp, P - knn/n, delta_embedding)
        scalings = [None] * len(names)
        if expression_scaling:
            # hi_dim @ (P - knn/n).T (:1716, and :1728 for the control) and the cosine projection of :1717 / :1729.  One launch
            # (vcy_embedding_scaling): groups of schedule-adjacent cells gather the rows of their common neighbours once, the
            # (genes, cells) estimates stay in registers.  Lists wider than it sorts in one workgroup take the two-step route.
            dS_r = self.dev(names[1][1]) if len(names) == 2 else None
            cos = ops.embedding_scaling(hi, self.dev(names[0][1]), neigh, parts[0][1], dS_r, parts[1][1] if len(names) == 2 else None, order=order,
                                        validate=False)
            if cos is None:
                indptr = torch.arange(0, (neigh.shape[0] + 1) * n, n, dtype=torch.int64, device=dev)
                if len(names) == 2:
                    estims = ops.knn_pool_w2(hi, indptr, neigh.reshape(-1), parts[0][1].reshape(-1), parts[1][1].reshape(-1), validate=False, order=order)
                else:
                    estims = (ops.knn_pool(hi, indptr, neigh.reshape(-1), parts[0][1].reshape(-1), validate=False, order=order),)
                cos = [ops.row_cosproj(self.dev(dS_name), estims[i]) for i, (_, dS_name) in enumerate(names)]                # :1717
                del estims
            for i in range(len(names)):
                scalings[i] = torch.clamp(cos[i] / scaling_penalty, 0, 1)                              # NaN stays NaN, like np.clip
        res = [(tp, de if sc is None else de * sc[:, None], sc) for (tp, _, de), sc in zip(parts, scalings)]

        tp, de, sc = res[0]
        self._tp, self._tp_ixs = tp, neigh
        self.delta_embedding = de.cpu().numpy()
        if sc is not None:
            self.scaling = sc.cpu().numpy()
        if len(res) == 2:
            tp, de, sc = res[1]
            self._tp_random = tp
            self.delta_embedding_random = de.cpu().numpy()
            if sc is not None:
                self.scaling_rndm = sc.cpu().numpy()

    # ------------------------------------------------------------------ consumers of the path ("next" rows)
    def calculate_grid_arrows(self, embed: str = "embedding", smooth: float = 0.5, steps: Tuple = (40, 40), n_neighbors: int = 100,
                              n_jobs: int = 4) -> None:
        """analysis.py:1735-1816: Gaussian-kernel average of delta_embedding on a regular grid.  The neighbour search
        of the grid points among the cells runs on the device (vcy_knn_query); the (grid x n_neighbors) weighting that
        follows is a few hundred KB and stays in NumPy like the reference."""
        embedding = np.asarray(getattr(self, embed), dtype=np.float64)
        if not hasattr(self, f"delta_{embed}"):
            raise KeyError("This embedding does not have a delta_*")
        delta_embedding = np.asarray(getattr(self, f"delta_{embed}"))
        grs = []
        for dim_i in range(embedding.shape[1]):
            m, M = np.min(embedding[:, dim_i]), np.max(embedding[:, dim_i])
            m = m - 0.025 * np.abs(M - m)
            M = M + 0.025 * np.abs(M - m)                      # (uses the widened m, as the reference does)
            grs.append(np.linspace(m, M, steps[dim_i]))
        gridpoints_coordinates = np.vstack([i.flat for i in np.meshgrid(*grs)]).T
        neighs, dists = ops.knn_query(embedding, gridpoints_coordinates, n_neighbors)
        neighs, dists = neighs.cpu().numpy().astype(np.int64), dists.cpu().numpy()
        std = np.mean([(g[1] - g[0]) for g in grs])
        scale = smooth * std
        gaussian_w = np.exp(-0.5 * (dists / scale) ** 2) / (scale * np.sqrt(2 * np.pi))     # scipy.stats.norm.pdf
        self.total_p_mass = gaussian_w.sum(1)
        UZ = (delta_embedding[neighs] * gaussian_w[:, :, None]).sum(1) / np.maximum(1, self.total_p_mass)[:, None]
        magnitude = np.linalg.norm(UZ, axis=1)
        self.flow_embedding = embedding
        self.flow_grid = gridpoints_coordinates
        self.flow = UZ
        self.flow_norm = UZ / np.percentile(magnitude, 99.5)
        self.flow_norm_magnitude = np.linalg.norm(self.flow_norm, axis=1)
        if "_corr_random" in self.__dict__ and hasattr(self, f"delta_{embed}_random"):
            UZ_rndm = (np.asarray(getattr(self, f"delta_{embed}_random"))[neighs] * gaussian_w[:, :, None]).sum(1) / np.maximum(1, self.total_p_mass)[:, None]
            magnitude_rndm = np.linalg.norm(UZ, axis=1)        # (sic: the reference scales the control by the real magnitudes, :1812)
            self.flow_rndm = UZ_rndm
            self.flow_norm_rndm = UZ_rndm / np.percentile(magnitude_rndm, 99.5)
            self.flow_norm_magnitude_rndm = np.linalg.norm(self.flow_norm_rndm, axis=1)

    def filter_genes_good_fit(self, minR: float = 0.1, min_gamma: float = 0.01) -> None:
        """analysis.py:1262-1265."""
        return self.filter_genes_by_phase_portrait(minR2=minR, min_gamma=min_gamma, minCorr=None)

    def filter_genes_by_phase_portrait(self, minR2: float = 0.1, min_gamma: float = 0.01, minCorr: float = 0.1) -> None:
        """analysis.py:1267-1319: drop genes with a poor fit (R2), a small gamma or a weak spliced/unspliced correlation
        (per-gene Pearson r from one moments pass on the device); every gene-indexed attribute is subset."""
        keep = np.ones(self.gammas.shape, dtype=bool)
        if minR2 is not None:
            keep &= (np.sqrt(np.abs(self.R2)) * np.sign(self.R2)) > minR2
        if min_gamma is not None:
            keep &= self.gammas > min_gamma
        if minCorr is not None:
            mom = ops.gene_moments(self.dev("Ux_sz"), self.dev("Sx_sz"))        # x = Sx_sz, y = Ux_sz
            n = float(self.dev("Sx_sz").C)
            sx, sy, sxx, sxy, syy = mom
            corr = ((sxy - sx * sy / n) / torch.sqrt((sxx - sx * sx / n) * (syy - sy * sy / n))).cpu().numpy()
            with np.errstate(invalid="ignore"):
                keep &= corr > minCorr
        self.ra = {k: v[keep] for k, v in self.ra.items()}
        kd = torch.from_numpy(keep)
        for name in ("U", "U_sz", "U_norm", "Ux", "Ux_sz", "Ux_norm", "S", "S_sz", "S_norm", "Sx", "Sx_sz", "Sx_norm"):
            if name in self._dev:
                self._set_dev(name, ops.select_genes(self._dev[name], kd))
        self.__dict__.get("_counts", {}).clear()                     # count-layer fast path no longer matches the gene set
        for name in ("gammas", "q", "R2"):
            if hasattr(self, name):
                setattr(self, name, getattr(self, name)[keep])

    # ------------------------------------------------------------------ stage F
    def prepare_markov(self, sigma_D: np.ndarray, sigma_W: np.ndarray, direction: str = "forward", cells_ixs: np.ndarray = None) -> None:
        """analysis.py:1818-1863.  cells_ixs restricts the chain to a subset of the cells (rows and columns of the
        transition probabilities and the embedding distances, as the reference slices its dense matrices)."""
        if direction not in ("forward", "backwards"):
            raise NotImplementedError(f"{direction} is not an implemented direction")
        embedding = np.asarray(self.embedding, dtype=np.float64)
        if cells_ixs is None:
            # all cells: the CSR of transition_prob is the compact (C, n) layout itself, its transpose one device sort
            tp, ixs = self._tp.double().contiguous(), self._tp_ixs.to(torch.int64).contiguous()
            C, n = tp.shape
            if direction == "forward":
                indptr, indices, data = torch.arange(0, C * n + 1, n, device=tp.device), ixs.ravel(), tp.ravel()
            else:
                rows, cols = ixs.ravel(), torch.arange(C, device=tp.device).repeat_interleave(n)
                order = torch.argsort(rows * C + cols)
                indptr = torch.zeros(C + 1, dtype=torch.int64, device=tp.device)
                indptr[1:] = torch.cumsum(torch.bincount(rows, minlength=C), 0)
                indices, data = cols[order], tp.ravel()[order]
        else:
            tp, ixs = self._tp.double().cpu().numpy(), self._tp_ixs.cpu().numpy().astype(np.int64)
            C, n = tp.shape
            P = sparse.csr_matrix((tp.ravel(), ixs.ravel(), np.arange(0, C * n + 1, n)), shape=(C, C))
            cells_ixs = np.asarray(cells_ixs)
            P = sparse.csr_matrix(P[cells_ixs, :][:, cells_ixs])
            embedding = embedding[cells_ixs, :]
            if direction == "backwards":
                P = sparse.csr_matrix(P.T)
            P.sort_indices()
            indptr, indices, data = P.indptr, P.indices, P.data
        # the chain is kept in factored form (sparse part + Gaussian of the embedding distance evaluated on the fly, in the
        # facade's storage type): no (n, n) matrix - 20 GB in fp64 at 50 000 cells - unless `tr` is asked for
        self._tr_dev = ops.prepare_markov_factored(indptr, indices, data, embedding, sigma_D, sigma_W, compute_dtype=self._dtype)

    def run_markov(self, starting_p: np.ndarray = None, n_steps: int = 2500, mode: str = "time_evolution") -> None:
        """analysis.py:1865-1887."""
        tr = self._tr_dev
        if starting_p is None:
            starting_p = np.ones(tr.shape[0]) / tr.shape[0]
        self.diffused = Diffusion().diffuse(starting_p, tr, n_steps=n_steps, mode=mode)[0]

    # ------------------------------------------------------------------ bookkeeping kept from the reference
    def to_hdf5(self, filename: str, **kwargs) -> None:
        """analysis.py:76-94: checkpoint of the whole object through serialization.dump_hdf5 (keyword arguments are handed on:
        data_compression, chunks, noarray_compression, pickle_protocol, exclude_attributes)."""
        from .serialization import dump_hdf5
        if "exclude" in kwargs:                                       # round-3 spelling
            kwargs["exclude_attributes"] = kwargs.pop("exclude")
        kwargs.setdefault("data_compression", 0)                      # (the reference's default of 7 costs minutes at 50 000 x 30 000)
        dump_hdf5(self, filename, **kwargs)

    def _export_state(self, exclude) -> Dict[str, Any]:
        """What a checkpoint holds for this object, as {attribute name: host value} (serialization.dump_hdf5 asks for it): the
        device matrices downloaded under their names as the reference's (genes, cells) float64 arrays, the pooling graph's scipy
        containers (assembled now if they still live on the device only), the plain attributes, and - where the reference
        persists dense (cells, cells) corrcoef / transition_prob, 20 GB each at 50 000 cells - the compact neighbour-list
        state under public names with the settings of the velocity chain."""
        out: Dict[str, Any] = {}
        for name in self._dev:
            if name not in exclude:
                out[name] = np.ascontiguousarray(getattr(self, name))
        for lazy_name in _LAZY_GRAPH:
            if lazy_name not in exclude:
                getattr(self, lazy_name, None)
        items = dict(self.__dict__)
        if "_neigh" in items and "embedding_knn" not in exclude:
            out["embedding_knn"] = self.embedding_knn                 # assembled on demand; a plain attribute in the reference
        for key, pub in _COMPACT_STATE.items():
            if key in items and pub not in exclude:
                out[pub] = items[key].cpu().numpy()
        chain = {k: items[k] for k in _CHAIN_SETTINGS if k in items}
        if chain:
            out["chain_settings"] = chain
        for name, val in items.items():
            if name.startswith("_") or name in exclude or isinstance(val, torch.Tensor):
                continue                                              # device-side caches are rebuilt on demand
            out[name] = val
        return out

    def _import_state(self) -> None:
        """After serialization.load_hdf5 has set the attributes of a checkpoint: rebuild the device-side state."""
        _restore_device_state(self)

    def reload_raw(self, substitute: bool = False) -> None:
        """analysis.py:2314-2342: read the layers of ``loom_filepath`` again, either over S, U, A, ca, ra (substitute) or
        next to them as ``raw_*``."""
        from .loom_io import read_loom
        layers, ca, ra = read_loom(self.loom_filepath)
        if substitute:
            self._init_layers(layers["spliced"], layers["unspliced"], layers.get("ambiguous"), ca, ra)
        else:
            self.raw_S, self.raw_U, self.raw_A = layers["spliced"], layers["unspliced"], layers.get("ambiguous")
            self.raw_initial_cell_size, self.raw_initial_Ucell_size = self.raw_S.sum(0), self.raw_U.sum(0)
            self.raw_ca, self.raw_ra = dict(ca), dict(ra)


def ixs_thatsort_a2b(a: np.ndarray, b: np.ndarray, check_content: bool = True) -> np.ndarray:
    """analysis.py:2345-2350: indices that put ``a`` in the order of ``b`` (same content)."""
    if check_content:
        assert len(np.intersect1d(a, b)) == len(a), f"The two arrays are not matching"
    return np.argsort(a)[np.argsort(np.argsort(b))]


def scale_to_match_median(sparse_matrix: sparse.csr_matrix, genes_total: np.ndarray) -> sparse.csc_matrix:
    """analysis.py:2392-2447: every stored weight of row i is scaled by min(1, median(t) / t) with t = the totals of the
    row's stored columns (returned with the csr index arrays reinterpreted as csc, exactly as the reference does)."""
    m = sparse.csr_matrix(sparse_matrix)
    genes_total = np.asarray(genes_total, dtype=np.float64)
    new = np.zeros(m.data.shape)
    for i in range(genes_total.shape[0]):
        lo, hi = m.indptr[i], m.indptr[i + 1]
        t = genes_total[m.indices[lo:hi]]
        new[lo:hi] = np.minimum(1, np.median(t) / t) * m.data[lo:hi]
    return sparse.csc_matrix((new, m.indices, m.indptr), shape=m.shape, copy=True)


def _scale_to_match_median(data: np.ndarray, indices: np.ndarray, indptr: np.ndarray, genes_total: np.ndarray) -> np.ndarray:
    """analysis.py:2392-2404: the loop of scale_to_match_median on the raw arrays of a sparse matrix."""
    new_data = np.zeros(data.shape)
    for i in range(genes_total.shape[0]):
        t = genes_total[indices[indptr[i]:indptr[i + 1]]]
        new_data[indptr[i]:indptr[i + 1]] = np.minimum(1, np.median(t) / t) * data[indptr[i]:indptr[i + 1]]
    return new_data


def numba_random_seed(value: int) -> None:
    """analysis.py:2407-2410: the reference seeds numba's private copy of numpy's legacy generator; there is no numba here, the
    module-level helpers below draw from numpy's global generator, which this seeds."""
    np.random.seed(value)


def permute_rows_nsign(A: np.ndarray) -> None:
    """analysis.py:2413-2420: in place, shuffle every row of A and flip the sign of its entries at random (host helper kept under
    the reference's name; estimate_transition_prob itself permutes on the device, `_permute_rows_nsign`).  Same draws as the
    reference's loop run without numba: np.random.shuffle, then np.random.choice([+1, -1]) per row."""
    plmi = np.array([+1, -1])
    for i in range(A.shape[0]):
        np.random.shuffle(A[i, :])
        A[i, :] = A[i, :] * np.random.choice(plmi, size=A.shape[1])


def _fill_diagonal_zero(m: torch.Tensor) -> None:
    m.diagonal().zero_()


def _permute_rows_nsign(dS: CellMatrix, seed: int, scratch=None) -> CellMatrix:
    """analysis.py:2407-2420: per gene, shuffle the values across cells and flip signs at random (the reference uses numba's RNG
    stream: statistical parity only).  On the device, ops.permute_rows_nsign."""
    return ops.permute_rows_nsign(dS, seed, scratch=scratch)


def gaussian_kernel(X: np.ndarray, mu: float = 0, sigma: float = 1) -> np.ndarray:
    """analysis.py:2449-2451."""
    return np.exp(-(X - mu)**2 / (2 * sigma**2)) / np.sqrt(2 * np.pi * sigma**2)


_COMPACT_STATE = {"_neigh": "embedding_knn_indices", "_corr": "corrcoef_compact", "_corr_random": "corrcoef_random_compact",
                  "_tp": "transition_prob_compact", "_tp_random": "transition_prob_random_compact", "_tp_ixs": "transition_prob_indices"}
_CHAIN_SETTINGS = ("_which_gamma", "_which_offset", "_vel_eps", "_assumption", "_shift_dt", "_extrap_dt", "_clip")


def _restore_device_state(vlm: "VelocytoLoom") -> None:
    """After a checkpoint has been read: rebuild the device-side state the methods downstream of estimate_transition_prob
    use.  Checkpoints of this package carry it in compact form; a checkpoint written by the reference carries the dense
    corrcoef / transition_prob and the embedding_knn graph, which are gathered back into neighbour lists here."""
    st = vlm.__dict__
    dev = ops.require_gpu()
    dt = vlm._dtype
    for k, v in (st.pop("chain_settings", None) or {}).items():
        st[k] = v
    for key, pub in _COMPACT_STATE.items():
        if pub in st:
            a = st.pop(pub)
            st[key] = torch.as_tensor(a).to(dev).to(torch.int32 if key in ("_neigh", "_tp_ixs") else dt).contiguous()
    if "_neigh" not in st and "embedding_knn" in st and sparse.issparse(st["embedding_knn"]):
        knn = sparse.csr_matrix(st["embedding_knn"])
        knn.sort_indices()
        lens = np.diff(knn.indptr)
        if lens.size and (lens == lens[0]).all():
            neigh = knn.indices.reshape(knn.shape[0], int(lens[0])).astype(np.int32)
            st["_neigh"] = torch.as_tensor(neigh).to(dev)
            rows = np.arange(knn.shape[0])[:, None]
            for dense, key in (("corrcoef", "_corr"), ("corrcoef_random", "_corr_random"), ("transition_prob", "_tp"), ("transition_prob_random", "_tp_random")):
                if dense in st and isinstance(st[dense], np.ndarray) and st[dense].shape == knn.shape:
                    st[key] = torch.as_tensor(np.ascontiguousarray(st.pop(dense)[rows, neigh])).to(dev).to(dt)
            if "_tp" in st:
                st["_tp_ixs"] = st["_neigh"]
    st.pop("embedding_knn", None)                                     # assembled from _neigh when read


def load_velocyto_hdf5(filename: str, dtype=None, obj_class: type = None) -> VelocytoLoom:
    """analysis.py:2454-2470: rebuild a VelocytoLoom from a checkpoint written by `to_hdf5` (or by the reference's dump_hdf5:
    same layout) through serialization.load_hdf5."""
    from .serialization import load_hdf5
    return load_hdf5(filename, obj_class=obj_class, dtype=dtype)
