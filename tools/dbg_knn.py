import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np, torch
import velocyto_amd
from velocyto_amd import ops
import oracle
C, P, k = 257, 2, 5
rng = np.random.default_rng(C * 7 + P * 3 + k)
X = np.round(rng.normal(size=(C, P)), 1)
idx, dist = ops.knn_search(X, k, include_self=False)
od, oi = oracle.knn_search(X, k, include_self=False)
idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
bad = np.where((idx != oi).any(1))[0]
print("bad rows", len(bad), bad[:10])
for r in bad[:4]:
    print(r, "got", idx[r], dist[r]); print("   want", oi[r], od[r])
    d2 = ((X - X[r]) ** 2).sum(1); d2[r] = np.inf
    o = np.lexsort((np.arange(C), d2))[:10]
    print("   brute", o, np.sqrt(d2[o]))
r = 21
for j in (72, 207):
    df = X[r] - X[j]
    print(j, X[r], X[j], df, repr(float((df * df).sum())), repr(float(df[0] * df[0] + df[1] * df[1])), repr(float(np.sum(df**2))))
d2 = ((X[r][None, None, :] - X[None, :, :]) ** 2).sum(-1)[0]
print(repr(d2[72]), repr(d2[207]))
