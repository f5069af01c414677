"""Kernel-level call surface of ``velocyto/speedboosted.pyx`` (speedboosted.pyx:542-610): the six ``_colDeltaCor*``
entry points with the Cython module's positional signatures and ownership rules - the caller allocates ``rm`` (cells x
cells, float64) and the result is ACCUMULATED into it (``rm[c, i] += ...``, speedboosted.pyx:78, 336); nothing is
returned.  Behind them are the same HIP kernels ``estimation.colDeltaCor*`` use (float64 storage here, like the
typed memoryviews ``double[:, ::1]`` of the reference).

Differences that are fixes, not omissions: ``e`` / ``d`` / ``ixs`` may be in any memory order (the reference raises
"ndarray is not C-contiguous"), and ``num_threads`` is accepted and ignored.
"""
from __future__ import annotations

import numpy as np

from . import estimation, ops

__all__ = ["_colDeltaCor", "_colDeltaCorSqrt", "_colDeltaCorLog10", "_colDeltaCorpartial", "_colDeltaCorSqrtpartial",
           "_colDeltaCorLog10partial"]


def _check_rm(rm: np.ndarray, C: int) -> None:
    if not isinstance(rm, np.ndarray) or rm.dtype != np.float64 or rm.shape != (C, C):
        raise ValueError(f"rm must be a float64 ndarray of shape {(C, C)} (the caller's zero-initialised output)")


def _full(e, d, rm, transform, psc) -> None:
    C = np.shape(e)[1]
    _check_rm(rm, C)
    rm += estimation._full(e, d, transform, psc, "float64")


def _partial(e, d, rm, ixs, transform, psc) -> None:
    C = np.shape(e)[1]
    _check_rm(rm, C)
    rm += estimation._partial(e, d, ixs, transform, psc, "float64")


def _colDeltaCor(e, d, rm, num_threads=None) -> None:
    """speedboosted.pyx:542-550 over x_colDeltaCor (:13-87)."""
    _full(e, d, rm, ops.LINEAR, 0.0)


def _colDeltaCorSqrt(e, d, rm, num_threads=None, psc: float = 0.0) -> None:
    """speedboosted.pyx:552-561 over x_colDeltaCorSqrt (:93-172)."""
    _full(e, d, rm, ops.SQRT, psc)


def _colDeltaCorLog10(e, d, rm, num_threads=None, psc: float = 1.0) -> None:
    """speedboosted.pyx:563-572 over x_colDeltaCorLog10 (:178-257)."""
    _full(e, d, rm, ops.LOG10, psc)


def _colDeltaCorpartial(e, d, rm, ixs, num_threads=None) -> None:
    """speedboosted.pyx:574-584 over x_colDeltaCorpartial (:263-346)."""
    _partial(e, d, rm, ixs, ops.LINEAR, 0.0)


def _colDeltaCorSqrtpartial(e, d, rm, ixs, num_threads=None, psc: float = 0.0) -> None:
    """speedboosted.pyx:586-597 over x_colDeltaCorSqrtpartial (:352-443)."""
    _partial(e, d, rm, ixs, ops.SQRT, psc)


def _colDeltaCorLog10partial(e, d, rm, ixs, num_threads=None, psc: float = 1.0) -> None:
    """speedboosted.pyx:599-610 over x_colDeltaCorLog10partial (:449-538)."""
    _partial(e, d, rm, ixs, ops.LOG10, psc)
