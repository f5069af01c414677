"""CSR pooling alone (vcy_knn_pool_csr) on a synthetic atlas block: the pass-1 pooling of both layers, against the dense uint8 kernel
on the densified layers of the same cells."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import velocyto_amd
from velocyto_amd import ops, atlas
dev = ops.require_gpu()
C, G, k = int(os.environ.get("C", 100000)), 30000, 30
cS, cU, totS, totU, pcs, emb = atlas.synth_atlas(C, G, 30, dev, density=0.08)
fS, fU = atlas.size_factors(totS, totU, C)
path = atlas.AtlasPath(cS, cU, fS, fU, pcs, emb, k=k, n_neighbors=500, sampled_fraction=0.5, block_cells=C)
o1, o2 = ops.CellMatrix.empty(C, G, torch.float32), ops.CellMatrix.empty(C, G, torch.float32)
def timeit(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def csr():
    path._pool(path.cS, path.fS, slice(0, C), o1); path._pool(path.cU, path.fU, slice(0, C), o2)
t = timeit(csr)
print(f"CSR pooling, {C} cells x 2 layers: {t:.2f} ms = {t / C * 1e3:.3f} us per cell", flush=True)
if not os.environ.get("SKIP_DENSE"):
    dS, dU = path.cS.to_dense(), path.cU.to_dense()
    ptr = torch.arange(0, (C + 1) * (k + 1), k + 1, device=dev, dtype=torch.int64)
    gi, gw = path.g_idx[:C].reshape(-1), path.g_w[:C].reshape(-1)
    td = timeit(lambda: ops.knn_pool_counts(dS, dU, path.fS, path.fU, ptr, gi, gw, dtype=torch.float32, out=o1, out2=o2, C_out=C, validate=False))
    print(f"dense uint8 pooling of the same cells: {td:.2f} ms; CSR / dense = {t / td:.2f}")
