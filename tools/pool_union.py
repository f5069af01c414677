#!/usr/bin/env python3
"""Stage A, the question of SURVEY 8 / north_star ("MFMA for the adjacency x count product where it is a genuine dense block contraction"):
how many DISTINCT count rows do groups of 8 / 16 cells that are adjacent in the pooling schedule gather (each cell: itself + its k = 30
nearest neighbours in the 30-d PCA space, neighbors.py:363-376)?  The union of a group is what an LDS-staged (or block-dense) pooling
kernel would read once per group; 31 x group size is what the gather kernel reads.  cfg3: 50 000 cells (the genes do not matter here).
Writes the distribution for the bench's own schedule (Hilbert curve over the two leading PCs) and for k-d leaf orders over more PCs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import velocyto_amd  # noqa: F401
from velocyto_amd import ops
import bench
dev = ops.require_gpu()
C, k = 50000, 30
cS, cU, fS, fU, pcs = bench.synth_counts(C, 3000, 30, dev)
idx, _ = ops.knn_search(pcs, k)
P, idxn = pcs.cpu().numpy(), idx.cpu().numpy()


def kd_order(X, dims, leaf):
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


def report(name, order):
    nb = np.concatenate([order[:, None], idxn[order]], 1)
    for T in (8, 16, 32):
        n = (C // T) * T
        grp = nb[:n].reshape(-1, T * (k + 1))
        u = np.array([len(np.unique(r)) for r in grp])
        refs = T * (k + 1)
        q = np.percentile(u, [5, 25, 50, 75, 95])
        print(f"{name:42s} group {T:2d}: gathers {refs:4d}  union median {q[2]:6.1f}  (5 % {q[0]:5.0f}, 25 % {q[1]:5.0f}, 75 % {q[3]:5.0f}, 95 % {q[4]:5.0f}, max {u.max():4d})"
              f"  = {q[2] / (k + 1):5.2f} x 31   sharing {refs / u.mean():4.2f} x   block density {refs / (T * u.mean()):5.3f}", flush=True)


report("Hilbert curve, 2 leading PCs (the bench's)", ops.hilbert_order(pcs[:, :2].contiguous()).cpu().numpy())
report("Morton curve, 3 leading PCs", ops.morton_order(pcs, 3).cpu().numpy())
for dims in (4, 8, 30):
    report(f"k-d leaves of 16 cells over {dims} PCs", kd_order(P, dims, 16))
print("""reading: `union median / 31` is the judge's criterion (stage the union through LDS if the median union of a group is <= 4 x 31 = 124 rows);
`sharing` = gathers / union = the factor by which LDS staging would cut the L2 -> CU gather traffic; `block density` = the fill of the
(group x union) block of the adjacency if it were multiplied as a dense block (MFMA): every other product is a multiplication by zero.""")
