"""Pooling variants on the default bench workload: uint16 / uint8 count layers, both layers in one launch or one each, slab widths."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import velocyto_amd
from velocyto_amd import ops
import bench
dev = ops.require_gpu()
DT = torch.float64 if os.environ.get("DTYPE", "f32") == "f64" else torch.float32      # storage type of the pooled matrices
print("pooled matrices:", DT)
C, G, k = 50000, 30000, 30
cS8, cU8, fS, fU, pcs = bench.synth_counts(C, G, 30, dev)
widen = lambda m: ops.CountMatrix(m.t.to(torch.int16), m.G) if m.t.dtype == torch.uint8 else m
cS16, cU16 = widen(cS8), widen(cU8)
space = pcs[:, :30].contiguous()
idx, dist_ = ops.knn_search(space, k)
wrow = torch.cat([torch.ones((C, 1), device=dev), (dist_ > 0).float()], 1)
wrow = (wrow / wrow.sum(1, keepdim=True)).contiguous()
indices = torch.cat([torch.arange(C, device=dev, dtype=torch.int32)[:, None], idx], 1).contiguous()
indptr = torch.arange(0, (C + 1) * (k + 1), k + 1, device=dev, dtype=torch.int64)
order = ops.morton_order(space, 3)
o1, o2 = ops.CellMatrix.empty(C, G, DT), ops.CellMatrix.empty(C, G, DT)
wrow = wrow.to(DT)
def timeit(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for name, a, b in (("u16", cS16, cU16), ("u8", cS8, cU8)):
    if a.t.dtype == torch.int16 and name == "u8":
        print("layers do not fit uint8"); continue
    for slab in (512, 1024, 2048, 4096):
        dual = timeit(lambda: ops.knn_pool_counts(a, b, fS, fU, indptr, indices, wrow, dtype=DT, out=o1, out2=o2, validate=False, order=order, slab_genes=slab))
        def two():
            ops.knn_pool_counts(a, None, fS, None, indptr, indices, wrow, dtype=DT, out=o1, validate=False, order=order, slab_genes=slab)
            ops.knn_pool_counts(b, None, fU, None, indptr, indices, wrow, dtype=torch.float32, out=o2, validate=False, order=order, slab_genes=slab)
        print(f"{name:4s} slab {slab:5d}: dual {dual:6.2f} ms   two launches {timeit(two):6.2f} ms", flush=True)
# where does the time go?  (a) every neighbour = the cell itself (all gathers hit L1/L2), (b) one neighbour only (output write + launch)
self_idx = torch.arange(C, device=dev, dtype=torch.int32)[:, None].repeat(1, k + 1).contiguous()
print("all-self neighbours: %.2f ms" % timeit(lambda: ops.knn_pool_counts(cS8, cU8, fS, fU, indptr, self_idx, wrow, dtype=DT, out=o1, out2=o2, validate=False, order=order)))
ip1 = torch.arange(0, C + 1, device=dev, dtype=torch.int64)
w1 = torch.ones(C, device=dev, dtype=DT)
print("one neighbour (self): %.2f ms" % timeit(lambda: ops.knn_pool_counts(cS8, cU8, fS, fU, ip1, self_idx[:, 0].contiguous(), w1, dtype=DT, out=o1, out2=o2, validate=False, order=order)))
near = (torch.arange(C, device=dev, dtype=torch.int64)[:, None] + torch.arange(-15, 16, device=dev)[None, :]).clamp(0, C - 1).to(torch.int32).contiguous()
print("31 index-adjacent neighbours, natural order: %.2f ms" % timeit(lambda: ops.knn_pool_counts(cS8, cU8, fS, fU, indptr, near, wrow, dtype=DT, out=o1, out2=o2, validate=False)))
print("real graph, natural order: %.2f ms" % timeit(lambda: ops.knn_pool_counts(cS8, cU8, fS, fU, indptr, indices, wrow, dtype=DT, out=o1, out2=o2, validate=False)))
if DT == torch.float64 or os.environ.get("SHORT"):
    sys.exit(0)
# float inputs (hand-edited S_sz / U_sz, size_norm=False on float layers): both matrices in one launch or one each
S32, U32 = cS8.to_float(torch.float32), cU8.to_float(torch.float32)
S32.t.mul_(fS[:, None].float()); U32.t.mul_(fU[:, None].float())
for slab in (0, 256, 512, 1024):
    dual = timeit(lambda: ops.knn_pool2(S32, U32, indptr, indices, wrow, out=o1, out2=o2, validate=False, order=order, slab_genes=slab))
    def two32():
        ops.knn_pool(S32, indptr, indices, wrow, out=o1, validate=False, order=order, slab_genes=slab)
        ops.knn_pool(U32, indptr, indices, wrow, out=o2, validate=False, order=order, slab_genes=slab)
    print(f"f32  slab {slab:5d}: dual {dual:6.2f} ms   two launches {timeit(two32):6.2f} ms", flush=True)
for slab in (512, 1024, 4096, 16384, 30016):
    t1 = timeit(lambda: ops.knn_pool_counts(cS8, None, fS, None, ip1, self_idx[:, 0].contiguous(), w1, dtype=DT, out=o1, validate=False, order=order, slab_genes=slab))
    t31 = timeit(lambda: ops.knn_pool_counts(cS8, None, fS, None, indptr, indices, wrow, dtype=DT, out=o1, validate=False, order=order, slab_genes=slab))
    print(f"u8 single layer, slab {slab:6d}: one neighbour {t1:5.2f} ms   31 neighbours {t31:5.2f} ms", flush=True)
# schedule orders for the pooling
for name, od in (("morton 3 PCs", ops.morton_order(space, 3)), ("morton 2 PCs", ops.morton_order(space, 2)), ("hilbert 2 PCs", ops.hilbert_order(space)), ("natural", None)):
    t = timeit(lambda: ops.knn_pool_counts(cS8, cU8, fS, fU, indptr, indices, wrow, dtype=DT, out=o1, out2=o2, validate=False, order=od))
    print(f"pool order {name:14s}: {t:5.2f} ms", flush=True)
