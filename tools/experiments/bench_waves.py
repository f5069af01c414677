#!/usr/bin/env python3
"""ALU efficiency of the stage-D pair body at 2 against 4 waves per SIMD, memory latency taken out: every cell of a 6-cell group gets the
SAME neighbour list, so a row carries six pairs and its load has six pair bodies to arrive.  Plain kernel (1024 threads) against the
pair-plan kernel run with a plan that pairs nothing (VCY_PLAN_THREADS threads)."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
a = types.SimpleNamespace(cells=int(os.environ.get("CELLS", 50000)), genes=int(os.environ.get("GENES", 30000)), k=30, pca_dims=30, n_neighbors=500,
                          sampled_fraction=0.5, curve="hilbert", order="embedding", exchange="halo", overlap=True, slab=0, fuse=True, literal_rule=True, counts="auto")
dev = torch.device("cuda", 0)
pipe = bench.Pipeline(a, dev, 0, 1, dtype=torch.float64)
ops = pipe.ops
gamma = pipe.step()
dmat = ops.velocity_chain(pipe.Sx_loc, pipe.Ux_loc, gamma, None, want=("dmat",), transform=ops.SQRT, psc=1e-10)["dmat"]
dm = ops.cell_moments(dmat)
C = a.cells
lead = (torch.arange(C, device=dev) // 6) * 6
shared = pipe.neigh[lead].contiguous()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(fn, n=3):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
out = torch.empty_like(pipe.corr_loc)
nothing = torch.full(shared.shape, -1, dtype=torch.int32, device=dev)
for name, ix, order in (("bench lists, Hilbert schedule", pipe.neigh_k, pipe.order), ("six cells per list, natural schedule", shared, None)):
    t1 = timed(lambda: ops.coldeltacor_partial(pipe.e_rows, dmat, ix, ops.SQRT, 1, 1e-10, order=order, out=out, validate=False))
    t2 = timed(lambda: ops.coldeltacor_partial_paired(pipe.e_rows, dmat, ix, ops.SQRT, 1, 1e-10, order=order, out=out, validate=False, plan=nothing, dm=dm))
    print(f"{name}: plain kernel {t1:.2f} ms; pair-plan kernel, nothing paired {t2:.2f} ms", flush=True)
