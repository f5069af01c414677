#!/usr/bin/env python3
"""One stage-D launch of the pair-plan kernel at the headline size for a counter pass (tools/pmc_kernel.sh): MODE=plan | empty | plain."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
a = types.SimpleNamespace(cells=int(os.environ.get("CELLS", 50000)), genes=int(os.environ.get("GENES", 30000)), k=30, pca_dims=30, n_neighbors=500,
                          sampled_fraction=0.5, curve="hilbert", order="embedding", exchange="halo", overlap=True, slab=0, fuse=True, literal_rule=True, counts="auto")
dev = torch.device("cuda", 0)
pipe = bench.Pipeline(a, dev, 0, 1, dtype=torch.float64 if os.environ.get("DTYPE", "f64") == "f64" else torch.float32)
ops = pipe.ops
gamma = pipe.step()
dmat = ops.velocity_chain(pipe.Sx_loc, pipe.Ux_loc, gamma, None, want=("dmat",), transform=ops.SQRT, psc=1e-10)["dmat"]
dm = ops.cell_moments(dmat)
mode = os.environ.get("MODE", "plan")
out = torch.empty_like(pipe.corr_loc)
for _ in range(int(os.environ.get("REPS", 2))):
    if mode == "plain":
        ops.coldeltacor_partial(pipe.e_rows, dmat, pipe.neigh_k, ops.SQRT, pipe.rules, 1e-10, order=pipe.order, out=out, validate=False)
    else:
        plan = ops.pair_plan(pipe.neigh_k, 0) if mode == "plan" else torch.full(pipe.neigh_k.shape, -1, dtype=torch.int32, device=dev)
        ops.coldeltacor_partial_paired(pipe.e_rows, dmat, pipe.neigh_k, ops.SQRT, pipe.rules, 1e-10, order=pipe.order, out=out, validate=False, plan=plan, dm=dm)
torch.cuda.synchronize()
