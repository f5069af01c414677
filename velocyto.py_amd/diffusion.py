"""Drop-in for ``velocyto/diffusion.py``: ``Diffusion.diffuse`` (diffusion.py:93-135).

The Markov step ``x <- x . tr`` runs on the device (``vcy_diffuse_step_dense`` for dense
matrices, ``vcy_diffuse_step_csc`` for scipy sparse ones); modes ``path_integral`` and
``time_evolution`` are what ``VelocytoLoom.run_markov`` uses (analysis.py:1887), ``map_trajectory`` /
``frontier`` reuse the same step.  The alternative transition-matrix builders and the stochastic
``trajectory`` mode are "next" rows.
"""
from __future__ import annotations

from typing import Any

import numpy as np
import torch
from scipy import sparse

from . import ops


class Diffusion:
    def __init__(self) -> None:
        pass

    def diffuse(self, x: np.ndarray, tr, n_steps: int = 10, mode: str = "path_integral") -> Any:
        """diffusion.py:93-135.  x: (n,) starting density (normalised to sum 1 like the reference),
        tr: (n, n) scipy sparse / numpy / torch right-stochastic matrix."""
        x = np.asarray(x, dtype=np.float64)
        x0 = x / x.sum()
        if not sparse.issparse(tr):
            tr = tr if isinstance(tr, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(tr, dtype=np.float64))
        if mode == "path_integral":
            _, acc = ops.diffuse(x0, tr, n_steps, accumulate=True)
            return acc.cpu().numpy()[None, :]                       # the reference returns a (1, n) matrix
        if mode == "time_evolution":
            xf, _ = ops.diffuse(x0, tr, n_steps, accumulate=False)
            return xf.cpu().numpy()[None, :]
        if mode in ("map_trajectory", "frontier"):
            cur = torch.from_numpy(x0).to(ops.require_gpu())
            result = [int(torch.argmax(cur))]
            for _ in range(n_steps):
                nxt, _ = ops.diffuse(cur, tr, 1, accumulate=False)
                result.append(int(torch.argmax(nxt)) if mode == "map_trajectory" else int(torch.argmax((nxt + 1) / (cur + 1))))
                cur = nxt
            return result
        raise NotImplementedError(f"mode={mode!r} is not implemented (SURVEY.md section 8f, 'next')")
