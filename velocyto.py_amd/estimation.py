"""Drop-in for ``velocyto/estimation.py`` (reference lines cited per function).

Same names, argument meaning and return types as the reference: matrices are passed in the
reference's ``(genes, cells)`` layout (numpy, any memory order - this also removes the reference's
F-order trap) or as device-resident ``ops.CellMatrix``; results come back as
numpy arrays of the reference's dtypes.  All arithmetic runs in the gfx950 kernels of
libvelocyto_hip.so; there is no CPU path.

``threads`` is accepted and ignored (the reference's rule, estimation.py:26-30, only sizes an
OpenMP pool).  Storage dtype: ``VELOCYTO_AMD_DTYPE`` (float64 default: the reference's arithmetic | float32: production mode) or ``dtype=``.
"""
from __future__ import annotations

from typing import Any, Optional, Tuple

import numpy as np
import torch

from . import ops
from .ops import CellMatrix

__all__ = ["colDeltaCor", "colDeltaCorpartial", "colDeltaCorLog10", "colDeltaCorLog10partial", "colDeltaCorSqrt",
           "colDeltaCorSqrtpartial", "fit_slope", "fit_slope_offset", "fit_slope_weighted", "fit_slope_weighted_offset",
           "clusters_stats"]


def _pair(emat, dmat, dtype) -> Tuple[CellMatrix, CellMatrix]:
    e = CellMatrix.from_genes_major(emat, dtype)
    d = CellMatrix.from_genes_major(dmat, dtype)
    if e.C != d.C or e.G != d.G:
        raise ValueError(f"emat {(e.G, e.C)} and dmat {(d.G, d.C)} must have the same shape")
    return e, d


def _dense_rows_block(C: int, itemsize: int) -> int:
    """Rows of the dense (C,C) result computed per launch so the device block stays <= ~4 GB."""
    return max(1, min(C, int(4e9 // max(C * itemsize, 1))))


def _full(emat, dmat, transform: int, psc: float, dtype) -> np.ndarray:
    e, d = _pair(emat, dmat, dtype)
    C = e.C
    out = np.zeros((C, C))                                   # estimation.py:31 -- caller-visible zeros, accumulated into
    blk = _dense_rows_block(C, e.t.element_size())
    for c0 in range(0, C, blk):
        n = min(blk, C - c0)
        rm = ops.coldeltacor_full(e, d, transform, psc, cell0=c0, C_out=n)
        out[c0:c0 + n] += rm.double().cpu().numpy()
    return out


def _partial(emat, dmat, ixs, transform: int, psc: float, dtype) -> np.ndarray:
    e, d = _pair(emat, dmat, dtype)
    C = e.C
    ixs = np.require(ixs, requirements="C").astype(np.intp)  # estimation.py:60
    if ixs.ndim != 2 or ixs.shape[0] != C:
        raise ValueError(f"ixs must be (ncells, nneighbours); got {ixs.shape} for {C} cells")
    comp = ops.coldeltacor_partial(e, d, ixs, transform, ops.partial_rules_for(e, transform, psc), psc)
    out = np.zeros((C, C))                                   # estimation.py:58
    blk = _dense_rows_block(C, comp.element_size())
    ix_dev = torch.from_numpy(ixs.astype(np.int32)).to(comp.device)
    for c0 in range(0, C, blk):                              # rm[c, ixs[c,n]] += ... (speedboosted.pyx:332-336)
        n = min(blk, C - c0)
        rm = ops.scatter_rows(comp[c0:c0 + n], ix_dev[c0:c0 + n], C)
        out[c0:c0 + n] += rm.double().cpu().numpy()
    return out


def colDeltaCor(emat, dmat, threads: int = None, dtype=None) -> np.ndarray:
    """estimation.py:11-33 over speedboosted.pyx:13-87."""
    return _full(emat, dmat, ops.LINEAR, 0.0, dtype)


def colDeltaCorpartial(emat, dmat, ixs, threads: int = None, dtype=None) -> np.ndarray:
    """estimation.py:36-62 over speedboosted.pyx:263-346."""
    return _partial(emat, dmat, ixs, ops.LINEAR, 0.0, dtype)


def colDeltaCorLog10(emat, dmat, threads: int = None, psc: float = 1.0, dtype=None) -> np.ndarray:
    """estimation.py:65-87 over speedboosted.pyx:178-257."""
    return _full(emat, dmat, ops.LOG10, psc, dtype)


def colDeltaCorLog10partial(emat, dmat, ixs, threads: int = None, psc: float = 1.0, dtype=None) -> np.ndarray:
    """estimation.py:90-116 over speedboosted.pyx:449-538."""
    return _partial(emat, dmat, ixs, ops.LOG10, psc, dtype)


def colDeltaCorSqrt(emat, dmat, threads: int = None, psc: float = 0.0, dtype=None) -> np.ndarray:
    """estimation.py:119-141 over speedboosted.pyx:93-172."""
    return _full(emat, dmat, ops.SQRT, psc, dtype)


def colDeltaCorSqrtpartial(emat, dmat, ixs, threads: int = None, psc: float = 0.0, dtype=None) -> np.ndarray:
    """estimation.py:144-170 over speedboosted.pyx:352-443."""
    return _partial(emat, dmat, ixs, ops.SQRT, psc, dtype)


# ------------------------------------------------------------------------------------------- fits
def fit_slope(Y, X, dtype=None) -> np.ndarray:
    """estimation.py:267-279 (+ _fit1_slope :173-188): float32 (genes,) slopes."""
    Yd, Xd = _pair(Y, X, dtype)
    return ops.fit_slope(Yd, Xd).cpu().numpy()


def _median_and_up_gamma(Yd: CellMatrix, Xd: CellMatrix) -> torch.Tensor:
    """The limit_gamma rule (estimation.py:199-205, 228-236) per gene:
    median(y) > median(x) ? max(1.5, percentile(y[x > percentile(x,90)], 10) / median(x[x > percentile(x,90)])) : 1.5"""
    med_y = ops.gene_quantiles(Yd, [50])[0]
    qx = ops.gene_quantiles(Xd, [50, 90])
    med_x, p90 = qx[0], qx[1]
    y10 = ops.gene_quantiles(Yd, [10], mask_src=Xd, mask_thr=p90, mask_mode=1)[0]
    xm = ops.gene_quantiles(Xd, [50], mask_src=Xd, mask_thr=p90, mask_mode=1)[0]
    up = torch.maximum(torch.full_like(y10, 1.5), y10 / xm)
    return torch.where(med_y > med_x, up, torch.full_like(up, 1.5))


def _fixperc_q(Yd: CellMatrix, Xd: CellMatrix) -> torch.Tensor:
    """m1 = percentile(y[x <= percentile(x, 1)], 50)   (estimation.py:222, 255)."""
    p1 = ops.gene_quantiles(Xd, [1])[0]
    return ops.gene_quantiles(Yd, [50], mask_src=Xd, mask_thr=p1, mask_mode=2)[0]


def fit_slope_offset(Y, X, fixperc_q: bool = False, dtype=None) -> Tuple[np.ndarray, np.ndarray]:
    """estimation.py:282-297 (+ _fit1_slope_offset :244-264)."""
    Yd, Xd = _pair(Y, X, dtype)
    if fixperc_q:
        q1 = _fixperc_q(Yd, Xd)
        m, q, _ = ops.fit_weighted(Yd, Xd, 2, fit_offset=False, lo_gamma=0.0, up_gamma_default=20.0, q_fixed=q1, want_R2=False)
    else:
        m, q, _ = ops.fit_weighted(Yd, Xd, 2, fit_offset=True, box_q=False, want_R2=False)
    return m.cpu().numpy(), q.cpu().numpy()


def fit_slope_weighted(Y, X, W, return_R2: bool = False, limit_gamma: bool = False, bounds: Tuple[float, float] = (0, 20),
                       dtype=None) -> Any:
    """estimation.py:300-334 (+ _fit1_slope_weighted :191-209).  As in the reference's row loop (:320)
    `bounds` is not forwarded: the search interval is (0, 20), or (1e-8, up_gamma) with limit_gamma."""
    Yd, Xd = _pair(Y, X, dtype)
    Wd = CellMatrix.from_genes_major(W, Yd.dtype)
    if limit_gamma:
        m, _, R2 = ops.fit_weighted(Yd, Xd, 0, W=Wd, fit_offset=False, lo_gamma=1e-8, up_gamma=_median_and_up_gamma(Yd, Xd))
    else:
        m, _, R2 = ops.fit_weighted(Yd, Xd, 0, W=Wd, fit_offset=False, lo_gamma=0.0, up_gamma_default=20.0)
    if return_R2:
        return m.cpu().numpy(), R2.cpu().numpy()
    return m.cpu().numpy()


def fit_slope_weighted_offset(Y, X, W, fixperc_q: bool = False, return_R2: bool = True, limit_gamma: bool = False, dtype=None) -> Any:
    """estimation.py:337-366 (+ _fit1_slope_weighted_offset :212-241).  The box-constrained weighted
    least-squares problem the reference hands to L-BFGS-B is solved exactly."""
    Yd, Xd = _pair(Y, X, dtype)
    Wd = CellMatrix.from_genes_major(W, Yd.dtype)
    m, q, R2 = _weighted_offset_device(Yd, Xd, 0, dict(W=Wd), fixperc_q, limit_gamma)
    if return_R2:
        return m.cpu().numpy(), q.cpu().numpy(), R2.cpu().numpy()
    return m.cpu().numpy(), q.cpu().numpy()


def _weighted_offset_device(Yd, Xd, weight_mode: int, wargs: dict, fixperc_q: bool, limit_gamma: bool):
    if fixperc_q:
        q1 = _fixperc_q(Yd, Xd)
        # (all-zero x -> (NaN, 0), all-zero y -> (0, 0) are applied inside the kernel; estimation.py:216-219)
        return ops.fit_weighted(Yd, Xd, weight_mode, fit_offset=False, lo_gamma=0.0, up_gamma_default=20.0, q_fixed=q1, **wargs)
    up = _median_and_up_gamma(Yd, Xd) if limit_gamma else None
    return ops.fit_weighted(Yd, Xd, weight_mode, fit_offset=True, box_q=True, lo_gamma=1e-8, up_gamma_default=20.0, up_gamma=up, **wargs)


def clusters_stats(U, S, clusters_uid: np.ndarray, cluster_ix: np.ndarray, size_limit: int = 40, dtype=None) -> Tuple[np.ndarray, np.ndarray]:
    """estimation.py:369-389: per-cluster gene averages; clusters of at most `size_limit` cells report the overall average.
    One masked streaming pass per cluster on the device (``vcy_gene_stats``).  U, S: (genes, cells) arrays or CellMatrix."""
    Ud, Sd = CellMatrix.from_genes_major(U, dtype), CellMatrix.from_genes_major(S, dtype)
    C, G = Sd.C, Sd.G
    cluster_ix = np.asarray(cluster_ix)
    U_avgs, S_avgs = np.zeros((G, len(clusters_uid))), np.zeros((G, len(clusters_uid)))
    overall = None
    for i, _ in enumerate(clusters_uid):
        sel = cluster_ix == i
        n = int(np.sum(sel))
        if n > size_limit:
            U_avgs[:, i] = ops.gene_stats(Ud, cell_mask=sel)[0].cpu().numpy() / n
            S_avgs[:, i] = ops.gene_stats(Sd, cell_mask=sel)[0].cpu().numpy() / n
        else:
            if overall is None:
                overall = (ops.gene_stats(Ud)[0].cpu().numpy() / C, ops.gene_stats(Sd)[0].cpu().numpy() / C)
            U_avgs[:, i], S_avgs[:, i] = overall
    return U_avgs, S_avgs


# --------------------------------------------------------------------------- one-gene forms (estimation.py:173-264)
def _one(v) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(v, dtype=np.float64)[None, :])


def _fit1_slope(y: np.ndarray, x: np.ndarray) -> float:
    """estimation.py:173-188 through the per-gene kernel (a (1, cells) problem)."""
    return float(fit_slope(_one(y), _one(x))[0])


def _fit1_slope_weighted(y: np.ndarray, x: np.ndarray, w: np.ndarray, limit_gamma: bool = False, bounds: Tuple[float, float] = (0, 20)) -> float:
    """estimation.py:191-209."""
    return float(np.asarray(fit_slope_weighted(_one(y), _one(x), _one(w), limit_gamma=limit_gamma, bounds=bounds)).ravel()[0])


def _fit1_slope_weighted_offset(y: np.ndarray, x: np.ndarray, w: np.ndarray, fixperc_q: bool = False, limit_gamma: bool = False) -> Tuple[float, float]:
    """estimation.py:212-241."""
    m, q = fit_slope_weighted_offset(_one(y), _one(x), _one(w), fixperc_q=fixperc_q, return_R2=False, limit_gamma=limit_gamma)[:2]
    return float(m[0]), float(q[0])


def _fit1_slope_offset(y: np.ndarray, x: np.ndarray, fixperc_q: bool = False) -> Tuple[float, float]:
    """estimation.py:244-264."""
    m, q = fit_slope_offset(_one(y), _one(x), fixperc_q=fixperc_q)
    return float(m[0]), float(q[0])
