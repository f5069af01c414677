import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, torch, numpy as np, bench
from velocyto_amd import ops
dev = ops.require_gpu()
C = 50000
_, _, pcs = bench.synth(C, 64, 30, dev)
emb = pcs[:, :2].double().contiguous()
gen = torch.Generator(device=dev).manual_seed(3)
m = 250
neigh, _ = ops.knn_search(emb.float(), m, include_self=False)
tp = torch.rand((C, m), generator=gen, device=dev, dtype=torch.float64) + 0.05
tp /= tp.sum(1, keepdim=True)
indptr = torch.arange(0, C * m + 1, m, device=dev)
x0 = torch.full((C,), 1.0 / C, dtype=torch.float64, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ext = float((emb.max(0).values - emb.min(0).values).max())
for sw, cull in ((4.0, False), (ext * 0.02, True), (ext * 0.005, True)):
    fac = ops.prepare_markov_factored(indptr, neigh.ravel(), tp.ravel(), emb, 2.0, sw, compute_dtype=torch.float64, cull=cull)
    ops.diffuse(x0, fac, 34, accumulate=False)
    e0.record(); x, _ = ops.diffuse(x0, fac, 400, accumulate=False); e1.record(); torch.cuda.synchronize()
    print(f"{sys.argv[1]:10s} f64 chain, {C} cells, sigma_W {sw:7.3f} (extent {ext:.1f}), culled={fac.cull is not None}: {e0.elapsed_time(e1) / 400:.3f} ms per step   sum {float(x.sum()):.15f}")
