"""Drop-in for ``velocyto/diffusion.py``: ``Diffusion.diffuse`` (diffusion.py:93-135).

The Markov step ``x <- x . tr`` runs on the device (``vcy_diffuse_step_dense`` for dense
matrices, ``vcy_diffuse_step_csc`` for scipy sparse ones); modes ``path_integral`` and
``time_evolution`` are what ``VelocytoLoom.run_markov`` uses (analysis.py:1887), ``map_trajectory`` /
``frontier`` reuse the same step, and the stochastic ``trajectory`` mode walks the chain on the host with numpy's
RNG like the reference.  The two alternative transition-matrix builders use the device kNN query.
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

    def compute_transition_matrix2(self, x0: np.ndarray, v: np.ndarray, sigma: float = 0.0, reverse: bool = False) -> sparse.csr_matrix:
        """diffusion.py:14-53: each cell is projected along its embedded velocity (x0 +/- v) and connected to the 20
        nearest cells of the projection with Gaussian weights of the distance; rows l1-normalised.  The neighbour
        search runs on the device (vcy_knn_query)."""
        x0, v = np.asarray(x0, dtype=np.float64), np.asarray(v, dtype=np.float64)
        n_cells, n_neighbors = x0.shape[0], 20
        x1 = x0 - v if reverse else x0 + v
        nearest, dists = ops.knn_query(x0, x1, n_neighbors)
        nearest, dists = nearest.cpu().numpy().astype(np.int64).ravel(), dists.cpu().numpy().ravel()
        probs = np.exp(-0.5 * (dists / sigma) ** 2) / (sigma * np.sqrt(2 * np.pi))            # scipy.stats.norm.pdf(dists, 0, sigma)
        cells = np.repeat(np.arange(n_cells), n_neighbors)
        return _l1_normalize_rows(sparse.coo_matrix((probs, (cells, nearest)), shape=(n_cells, n_cells)).tocsr())

    def compute_transition_matrix(self, knn: sparse.spmatrix, x: np.ndarray, v: np.ndarray, epsilon: float = 0.0, reverse: bool = False
                                  ) -> sparse.csr_matrix:
        """diffusion.py:55-91: transition probability along a kNN edge = scalar projection of the cell's velocity on the
        edge direction (+ epsilon, clipped at 0) / edge length; rows l1-normalised.  O(edges) index arithmetic, NumPy."""
        knn = sparse.coo_matrix(knn)
        v0, v1 = knn.row, knn.col
        x, v = np.asarray(x, dtype=np.float64), np.asarray(v, dtype=np.float64)
        uv = x[v1] - x[v0]
        norms = np.linalg.norm(uv, axis=1)
        with np.errstate(divide="ignore", invalid="ignore"):
            uv = uv / norms[:, None]
            scalar_projection = (v[v0] * uv).sum(1)
            if reverse:
                scalar_projection = -scalar_projection
            scalar_projection = np.clip(scalar_projection + epsilon, 0, None)
            p = scalar_projection * (1 / norms)
        return _l1_normalize_rows(sparse.coo_matrix((p, (v0, v1))).tocsr())

    def diffuse(self, x: np.ndarray, tr, n_steps: int = 10, mode: str = "path_integral") -> Any:
        """diffusion.py:93-135.  x: (n,) starting density (normalised to sum 1 like the reference),
        tr: (n, n) scipy sparse / numpy / torch right-stochastic matrix."""
        x = np.asarray(x, dtype=np.float64)
        x0 = x / x.sum()
        if isinstance(tr, ops.MarkovFactors):
            if mode == "trajectory":
                tr = tr.dense()                                     # the random walk reads single rows of tr
        elif not sparse.issparse(tr):
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
        if mode == "trajectory":
            # diffusion.py:121-135: a random walk, one node per step drawn from the current node's row of tr (l1-normalised),
            # with numpy's global RNG exactly as the reference (same stream for the same seed and probabilities)
            def row(node: int) -> np.ndarray:
                if sparse.issparse(tr):
                    return np.asarray(tr.getrow(node).todense()).ravel().astype(np.float64)
                return tr[node].detach().to(torch.float64).cpu().numpy()
            n = x.shape[0]
            node = np.random.choice(np.arange(n), p=x)
            trajectories = [node]
            for _ in range(n_steps):
                nxt = row(int(node))
                tot = np.abs(nxt).sum()
                if tot == 0:                                        # no way out: stay (normalize() leaves a zero row as it is)
                    nxt = np.zeros(n)
                    nxt[node] = 1.0
                else:
                    nxt = nxt / tot
                node = np.random.choice(np.arange(n), p=nxt)
                trajectories.append(node)
            return trajectories
        raise NotImplementedError(f"mode={mode!r} is not a mode of Diffusion.diffuse")


def _l1_normalize_rows(m: sparse.csr_matrix) -> sparse.csr_matrix:
    """sklearn.preprocessing.normalize(m, axis=1, norm="l1") (diffusion.py:52, 90): rows with zero norm are left as they are."""
    m = sparse.csr_matrix(m, dtype=np.float64)
    norms = np.asarray(abs(m).sum(1)).ravel()
    norms[norms == 0] = 1.0
    return sparse.diags(1.0 / norms) @ m
