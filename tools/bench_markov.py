"""Time the Markov steps of run_markov (factored chain: sparse part + Gauss transform of the embedding evaluated on the fly) with the
full and the culled transform, for kernels of different width against the embedding.
usage (GPU box): [C=50000 STEPS=200] python tools/bench_markov.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from velocyto_amd import ops
import bench

C, steps = int(os.environ.get("C", 50000)), int(os.environ.get("STEPS", 200))
dev = ops.require_gpu()
_, _, pcs = bench.synth(C, 64, 30, dev)
emb = pcs[:, :2].double().contiguous()
ext = float((emb.max(0).values - emb.min(0).values).max())
gen = torch.Generator(device=dev).manual_seed(3)
m = 250
neigh, _ = ops.knn_search(emb.float(), m, include_self=False)
tp = torch.rand((C, m), generator=gen, device=dev, dtype=torch.float64) + 0.05
tp /= tp.sum(1, keepdim=True)
indptr = torch.arange(0, C * m + 1, m, device=dev)
x0 = torch.full((C,), 1.0 / C, dtype=torch.float64, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
print(f"{C} cells, embedding extent {ext:.2f}, {steps} steps (hipGraph replay)")
for frac in (0.005, 0.01, 0.02, 0.05, 0.2):
    sw = ext * frac
    row = []
    res = []
    for cull in (False, True):
        fac = ops.prepare_markov_factored(indptr, neigh.ravel(), tp.ravel(), emb, 2 * sw, sw, compute_dtype=torch.float32, cull=cull)
        ops.diffuse(x0, fac, 34, accumulate=False)
        e0.record(); x, _ = ops.diffuse(x0, fac, steps, accumulate=False); e1.record(); torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / steps)
        res.append(x)
    # what an ideal culling would leave: pairs within the radius where the kernel is dropped (cut 48: exp2(-d^2 / (2 ln2 s^2)) ...)
    radius = (48.0 * 2.0 * sw * sw / 1.4426950408889634) ** 0.5
    samp = emb[:: max(1, C // 512)]
    inrange = float((torch.cdist(samp, emb) < radius).double().mean())
    # what the kernel's box tests leave: (256-target block, 32-source chunk) pairs within reach, and how unevenly they fall on workgroups
    es_s, _, boxes, cut, _ = fac.cull
    nc = (C + 31) // 32
    bx = boxes.view(torch.float32)
    clo, chi = bx[: nc * 2].view(nc, 2), bx[nc * 2: nc * 4].view(nc, 2)
    TB = 256
    nb = (C + TB - 1) // TB
    pad = torch.cat([es_s, es_s[-1:].expand(nb * TB - C, 2)])
    tlo, thi = pad.view(nb, TB, 2).amin(1), pad.view(nb, TB, 2).amax(1)
    gap = torch.maximum(torch.maximum(clo[None] - thi[:, None], tlo[:, None] - chi[None]), torch.zeros((), device=dev))
    near = (gap * gap).sum(-1) <= cut                                   # (nb, nc)
    nparts = min(64, nc)
    qper = (nc + nparts - 1) // nparts
    part_of_chunk = torch.arange(nc, device=dev) // qper
    load = torch.zeros((nb, nparts), device=dev).index_add_(1, part_of_chunk, near.float())
    kept, worst = float(near.float().mean()), float(load.max()) * 32
    auto = ops.prepare_markov_factored(indptr, neigh.ravel(), tp.ravel(), emb, 2 * sw, sw, compute_dtype=torch.float32).cull is not None
    rel = float(((res[0] - res[1]).abs() / res[0].abs().clamp_min(1e-300)).max())
    print(f"sigma_W = {frac:5.3f} x extent: full {row[0]:7.4f} ms/step, culled {row[1]:7.4f} ms/step ({row[0] / row[1]:5.1f}x), auto picks {'culled' if auto else 'full'}; pairs in range {inrange:.4f}, kept by the box tests {kept:.4f}, busiest workgroup {worst:.0f} sources (mean {float(load.mean()) * 32:.0f}); max rel diff after {steps} steps {rel:.2e}")
