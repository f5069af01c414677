"""Diagnostic: how many distinct kNN rows does a tile of T Morton-adjacent cells touch (pooling reuse potential)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import velocyto_amd
from velocyto_amd import ops
import bench
dev = ops.require_gpu()
C, G = 50000, 3000
S, U, pcs = bench.synth(C, G, 30, dev)
idx, _ = ops.knn_search(pcs, 30)
for dims in (2, 3):
    order = ops.morton_order(pcs, dims).long()
    nb = torch.cat([order[:, None].to(torch.int32), idx[order]], 1).cpu().numpy()
    for T in (8, 16, 32, 64):
        n = (C // T) * T
        grp = nb[:n].reshape(-1, T * nb.shape[1])
        distinct = np.array([len(np.unique(r)) for r in grp[::20]])
        print(f"morton dims={dims} tile={T:3d}: refs {T*31:5d} distinct {distinct.mean():7.1f} mult {T*31/distinct.mean():.2f}")

def kd_order(X, dims, leaf=16):
    """k-d tree leaf order over the first `dims` coordinates (recursive median split on the widest dimension)."""
    out = []
    def rec(ix):
        if len(ix) <= leaf:
            out.append(ix); return
        sub = X[ix][:, :dims]
        d = int(np.argmax(sub.max(0) - sub.min(0)))
        o = np.argsort(sub[:, d], kind="stable")
        h = len(ix) // 2
        rec(ix[o[:h]]); rec(ix[o[h:]])
    rec(np.arange(X.shape[0]))
    return np.concatenate(out)

P = pcs.cpu().numpy()
idxn = idx.cpu().numpy()
for dims in (4, 8, 16, 30):
    order = kd_order(P, dims)
    nb = np.concatenate([order[:, None], idxn[order]], 1)
    for T in (16, 32, 64):
        n = (C // T) * T
        grp = nb[:n].reshape(-1, T * nb.shape[1])
        distinct = np.array([len(np.unique(r)) for r in grp[::20]])
        print(f"kd dims={dims:2d} tile={T:3d}: refs {T*31:5d} distinct {distinct.mean():7.1f} mult {T*31/distinct.mean():.2f}")
