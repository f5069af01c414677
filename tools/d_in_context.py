#!/usr/bin/env python3
"""Why does stage D take longer inside the pass than launched alone?  The headline pipeline (f64, uint16 layers) timed four ways on one box:
whole steps; the fused stage-D launch alone, back to back; alone after a pooling launch; alone after an idle gap."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
a = types.SimpleNamespace(cells=50000, genes=30000, k=30, pca_dims=30, n_neighbors=500, sampled_fraction=0.5, curve="hilbert", order="embedding",
                          exchange="halo", overlap=True, slab=0, fuse=True, literal_rule=True, counts="u16")
dev = torch.device("cuda", 0)
pipe = bench.Pipeline(a, dev, 0, 1, dtype=torch.float64, counts="u16")
pipe.probe = None
ops = pipe.ops
for _ in range(2):
    pipe.step()
torch.cuda.synchronize()
gamma = pipe.last_gamma
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def d_only():
    ops.coldeltacor_partial_fused(pipe.e_rows, pipe.Ux_loc, gamma, None, pipe.neigh_k, ops.SQRT, pipe.rules, 1e-10, order=pipe.order, out=pipe.corr_loc, validate=False)
def timed_d(before=None, n=5):
    ts = []
    for _ in range(n):
        if before: before()
        e0.record(); d_only(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return np.mean(ts), np.min(ts)
pipe.d_ms = []
for _ in range(5):
    pipe.step(timed=True)
print(f"D inside whole steps:                 mean {np.mean(pipe.d_ms):.2f} ms  min {np.min(pipe.d_ms):.2f}")
print("D alone, back to back:                mean %.2f ms  min %.2f" % timed_d())
def pool():
    k = a.k
    idx, dist_ = ops.knn_search(pipe.space, k, include_self=False)
    conn = (dist_ > 0).to(pipe.dtype)
    wrow = torch.cat([torch.ones((a.cells, 1), device=dev, dtype=pipe.dtype), conn], 1); wrow = wrow / wrow.sum(1, keepdim=True)
    indices = torch.cat([torch.arange(a.cells, device=dev, dtype=torch.int32)[:, None], idx], 1)
    indptr = torch.arange(0, (a.cells + 1) * (k + 1), k + 1, device=dev, dtype=torch.int64)
    indices, wrow = ops.canonical_graph_rows(indices, wrow)
    ops.knn_pool_counts(pipe.cS, pipe.cU, pipe.fS, pipe.fU, indptr, indices, wrow, dtype=pipe.dtype, out=pipe.Sx_loc, out2=pipe.Ux_loc, validate=False, order=pipe.pool_order)
print("D alone, after kNN + pooling:         mean %.2f ms  min %.2f" % timed_d(pool))
def fit():
    ops.fit_slope_from_moments(ops.fit_slope_moments(pipe.Ux_loc, pipe.Sx_loc))
print("D alone, after fit_slope:             mean %.2f ms  min %.2f" % timed_d(fit))
print("D alone, after 0.3 s idle:            mean %.2f ms  min %.2f" % timed_d(lambda: time.sleep(0.3)))
print("D alone, back to back again:          mean %.2f ms  min %.2f" % timed_d())
dmat = ops.velocity_chain(pipe.Sx_loc, pipe.Ux_loc, gamma, None, want=("dmat",), transform=ops.SQRT, psc=1e-10)["dmat"]
def d_mat():
    ops.coldeltacor_partial(pipe.e_rows, dmat, pipe.neigh_k, ops.SQRT, pipe.rules, 1e-10, order=pipe.order, out=pipe.corr_loc, validate=False)
ts = []
for _ in range(5):
    e0.record(); d_mat(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print("D with a materialised d, back to back: mean %.2f ms  min %.2f" % (np.mean(ts), np.min(ts)))
