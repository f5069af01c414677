"""The linear all-pairs kernel (vcy_coldeltacor_full_linear + its repair launch) on POOLED matrices - the population the repair pass exists for:
every cell the kNN mean of 31 cells, neighbours share most of their pool.  Prints the launch time, the fraction of pairs the epilogue hands to
the repair pass (variance below 2^-10 of the terms it was expanded from; recomputed here with torch), and the time on unpooled data for scale."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from velocyto_amd import ops
dev = ops.require_gpu()
C, G = int(os.environ.get("C", 10000)), int(os.environ.get("G", 20000))
sys.argv = [sys.argv[0], "--no-extra", "--no-cpu-baseline", "--cells", str(C), "--genes", str(G)]
a = bench.parse()
pipe = bench.Pipeline(a, dev, 0, 1, dtype=torch.float64, counts=a.counts)
gamma = pipe.step()
Sx, Ux = pipe.Sx_loc, pipe.Ux_loc
d = ops.velocity_chain(Sx, Ux, gamma, None, want=("delta_S",))["delta_S"]          # the linear variant correlates with delta_S itself


def best(f, reps=3):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3


rm = torch.empty((C, C), dtype=torch.float64, device=dev)
t_pooled = best(lambda: ops.coldeltacor_full(Sx, d, ops.LINEAR, rm=rm, validate=False))
E = Sx.t[:, :G]
n = float(G)
Se, See = E.sum(1), (E * E).sum(1)
flagged = 0
for r0 in range(0, C, 2000):
    r1 = min(C, r0 + 2000)
    sAA = See[None, :] + See[r0:r1, None] - 2.0 * (E[r0:r1] @ E.T)
    sA = Se[None, :] - Se[r0:r1, None]
    va = sAA - sA * sA / n
    f = va < (See[None, :] + See[r0:r1, None]) / 1024.0
    f[torch.arange(r1 - r0, device=dev), torch.arange(r0, r1, device=dev)] = False
    flagged += int(f.sum())
raw = ops.CellMatrix(torch.rand((C, ops.padded_ld(G)), device=dev, dtype=torch.float64), G)
raw.t[:, G:] = 0
t_raw = best(lambda: ops.coldeltacor_full(raw, d, ops.LINEAR, rm=rm, validate=False))
# accuracy at scale on the pooled matrices: the matrix-core route (with its repair launch) against the element-wise kernel (raw fp64 moments of the differences)
mm = ops.coldeltacor_full(Sx, d, ops.LINEAR, validate=False)
ops.FULL_LINEAR_MFMA = False
ew = ops.coldeltacor_full(Sx, d, ops.LINEAR, validate=False)
ops.FULL_LINEAR_MFMA = True
both = torch.isfinite(mm) & torch.isfinite(ew)
print(f"pooled matrices, all {C * C} pairs: NaN pattern equal {bool(torch.equal(torch.isnan(mm), torch.isnan(ew)))}, max |matrix-core - element-wise| = {float((mm[both] - ew[both]).abs().max()):.2e}")
del mm, ew
fin = torch.isfinite(rm)
print(f"linear all-pairs kernel, {C} cells x {G} genes, f64: pooled Sx (k = 30) {t_pooled:.1f} ms with {flagged} of {C * (C - 1)} pairs ({flagged / (C * (C - 1)):.2e}) "
      f"re-evaluated by the repair launch; random matrix (nothing flagged) {t_raw:.1f} ms")
