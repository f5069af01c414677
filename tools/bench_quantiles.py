"""Time vcy_gene_quantiles (np.percentile over the cells of every gene; fit_gammas' weights and limits) on the GPU box.
usage: [C=50000 G=30000 DT=float32] python tools/bench_quantiles.py      (VCY_QUANTILES_REG=0: the row-re-reading kernel)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from velocyto_amd import ops

C, G = int(os.environ.get("C", 50000)), int(os.environ.get("G", 30000))
dt = getattr(torch, os.environ.get("DT", "float32"))
dev = ops.require_gpu()
g = torch.Generator(device=dev).manual_seed(1)
t = torch.empty((C, ops.padded_ld(G)), dtype=dt, device=dev)
t.copy_(torch.rand(t.shape, generator=g, device=dev, dtype=torch.float32) * (torch.rand(t.shape, generator=g, device=dev, dtype=torch.float32) < 0.4))
m = ops.CellMatrix(t, G)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for qs in ([98], [2, 98], [0, 2, 50, 98, 100]):
    best = 1e9
    for _ in range(4):
        e0.record(); out = ops.gene_quantiles(m, qs); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print(f"C={C} G={G} {dt} percentiles {qs}: {best:.2f} ms (transpose + selection)   checksum {float(out.sum()):.6f}")
