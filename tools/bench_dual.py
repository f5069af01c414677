"""Stage D with the randomised control: one dual launch against two single launches (50k x 30k, nrndm 250).
VERDICT r1 item 4: done = dual <= 1.15 x one single launch, outputs bit-identical to the two launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import velocyto_amd
from velocyto_amd import ops
import bench

dev = ops.require_gpu()
C, G = int(os.environ.get("C", 50000)), int(os.environ.get("G", 30000))
DT = torch.float64 if os.environ.get("DTYPE", "f32") == "f64" else torch.float32
S, U, pcs = bench.synth(C, G, 30, dev)
if DT == torch.float64:
    S = ops.CellMatrix(S.t.double(), G)
del U
emb = pcs[:, :2].contiguous()
neigh, _ = bench.sample_neighbors_device(emb, 500, 0.5, dev)
order = ops.hilbert_order(emb)
gen = torch.Generator(device=dev).manual_seed(3)
d = ops.CellMatrix(torch.randn(S.t.shape, generator=gen, device=dev).to(DT), G)
d2 = ops.CellMatrix(torch.randn(S.t.shape, generator=gen, device=dev).to(DT), G)
d.t[:, G:] = 0
d2.t[:, G:] = 0
o1 = torch.empty((C, neigh.shape[1]), dtype=DT, device=dev)
o2 = torch.empty_like(o1)
p1 = torch.empty_like(o1)
p2 = torch.empty_like(o1)


def timed(fn, n=3):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3


rules = ops.RULES_PARTIAL if os.environ.get("LITERAL") else ops.partial_rules_for(S, ops.SQRT, 1e-10)
print("storage:", os.environ.get("DTYPE", "f32"), " rule:", ops.RULE_NAMES[rules], " VCY_CDC_DUAL_F64 =", os.environ.get("VCY_CDC_DUAL_F64", "(default)"))
single = timed(lambda: ops.coldeltacor_partial(S, d, neigh, ops.SQRT, rules, 1e-10, order=order, out=o1, validate=False))
ops.coldeltacor_partial(S, d2, neigh, ops.SQRT, rules, 1e-10, order=order, out=o2, validate=False)
dual = timed(lambda: ops.coldeltacor_partial_dual(S, d, d2, neigh, ops.SQRT, rules, 1e-10, order=order, out=p1, out_rndm=p2, validate=False))
same = float(torch.nan_to_num(o1 - p1).abs().max()), float(torch.nan_to_num(o2 - p2).abs().max())
print(f"single launch {single:.2f} ms   two launches {2 * single:.2f} ms   dual launch {dual:.2f} ms   dual/single {dual / single:.3f}   max |dual - single| {same}")
