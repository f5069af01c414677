"""Micro-benchmark of the stage-D kernels alone (grouped vs one-cell-per-workgroup, per transform)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import velocyto_amd
from velocyto_amd import ops
import bench

dev = ops.require_gpu()
C, G = int(os.environ.get("C", 50000)), int(os.environ.get("G", 30000))
S, U, pcs = bench.synth(C, G, 30, dev)
emb = pcs[:, :2].contiguous()
neigh, _ = bench.sample_neighbors_device(emb, 500, 0.5, dev)
order = ops.morton_order(emb, 2)
if os.environ.get("NEIGH_RANGE"):   # diagnostic: all neighbour rows from a tiny set -> served by L2 (non-memory floor)
    nr = int(os.environ["NEIGH_RANGE"])
    neigh = torch.randint(0, nr, neigh.shape, device=dev, dtype=torch.int32)
d = ops.CellMatrix(torch.randn_like(S.t), G)
out = torch.empty((C, neigh.shape[1]), dtype=torch.float32, device=dev)
for tr, name in ((ops.SQRT, "sqrt"), (ops.LINEAR, "linear"), (ops.LOG10, "log10")):
    for od, oname in ((order, "morton"), (None, "natural")):
        ts = []
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ops.coldeltacor_partial(S, d, neigh, tr, ops.RULES_PARTIAL, 1e-10, order=od, out=out, validate=False)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"group={os.environ.get('VCY_CDC_GROUP','auto')} {name:7s} {oname:8s} {min(ts)*1e3:8.2f} ms")
