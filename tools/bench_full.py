"""Time the all-pairs (full) correlation kernels; the LINEAR variant three ways: the f64 matrix-core kernel (vcy_coldeltacor_full_linear),
the element-wise kernel, and - for scale only - the same algebra as two library GEMMs + eager elementwise ops (round 4's route, kept here)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import velocyto_amd
from velocyto_amd import ops
dev = ops.require_gpu()
C, G = int(os.environ.get("C", 10000)), int(os.environ.get("G", 20000))
dt_ = torch.float64 if os.environ.get("DTYPE", "f32") == "f64" else torch.float32
e = ops.CellMatrix(torch.rand((C, ops.padded_ld(G)), device=dev, dtype=dt_), G)
d = ops.CellMatrix(torch.randn((C, ops.padded_ld(G)), device=dev, dtype=dt_), G)
e.t[:, G:] = 0; d.t[:, G:] = 0


def timed(fn, n=2):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(n):
        t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return r, best


def library_route(block=4096):
    n = float(G)
    E = e.t[:, :G].double()
    Se, See = E.sum(1), (E * E).sum(1)
    rm = torch.empty((C, C), dtype=dt_, device=dev)
    for r0 in range(0, C, block):
        r1 = min(C, r0 + block)
        Eb, Db = E[r0:r1], d.t[r0:r1, :G].double()
        sb, sbb, sed = Db.sum(1), (Db * Db).sum(1), (Eb * Db).sum(1)
        sA = Se[None, :] - Se[r0:r1, None]
        sAA = See[None, :] + See[r0:r1, None] - 2.0 * (Eb @ E.T)
        sAb = Db @ E.T - sed[:, None]
        rm[r0:r1] = ((sAb - sA * sb[:, None] / n) / torch.sqrt((sAA - sA * sA / n) * (sbb - sb * sb / n)[:, None])).to(dt_)
    return rm


for tr, name in (() if os.environ.get("ONLY_LINEAR") else ((ops.SQRT, "sqrt"), (ops.LOG10, "log10"))):
    _, t = timed(lambda: ops.coldeltacor_full(e, d, tr, 1e-10), 1)
    print(f"full {name:7s} C={C} G={G} {dt_}: {t*1e3:8.1f} ms  {C*C*G/t/1e12:.2f} T pair-genes/s")
def clocked(fn, ms):
    pr = ops.ClockProbe(interval_ms=1.0)
    fn(); torch.cuda.synchronize()
    pr.start(0.8 * ms)
    fn(); torch.cuda.synchronize()
    return pr.ghz()


a, t1 = timed(lambda: ops.coldeltacor_full(e, d, ops.LINEAR))
flop = 2 * 2.0 * C * C * G
print(f"full linear, f64 matrix-core kernel  C={C} G={G} {dt_}: {t1*1e3:8.1f} ms  {flop/t1/1e12:.1f} Tflop/s = {flop/t1/78.6e12:.2f} of the f64 matrix peak")
ops.FULL_LINEAR_MFMA = False
b, t2 = timed(lambda: ops.coldeltacor_full(e, d, ops.LINEAR), 1)
ops.FULL_LINEAR_MFMA = True
print(f"full linear, element-wise kernel     C={C} G={G} {dt_}: {t2*1e3:8.1f} ms")
c, t3 = timed(library_route)
print(f"full linear, two library GEMMs + eager epilogue (round 4)      : {t3*1e3:8.1f} ms")
g1, g3 = clocked(lambda: ops.coldeltacor_full(e, d, ops.LINEAR), t1 * 1e3), clocked(library_route, t3 * 1e3)
print(f"shader clock while running (mean, min, max GHz): matrix-core kernel {g1[0]:.2f} {g1[1]:.2f} {g1[2]:.2f}; library route {g3[0]:.2f} {g3[1]:.2f} {g3[2]:.2f}"
      f"  -> the kernel runs at {flop/t1/(78.6e12*g1[0]/2.4):.2f} of the matrix peak AT ITS CLOCK")
off = ~torch.eye(C, dtype=torch.bool, device=dev)
ok = off & torch.isfinite(a) & torch.isfinite(b)
print(f"max |matrix-core - element-wise| = {float((a[ok] - b[ok]).abs().max()):.3e}; max |matrix-core - library route| = {float((a[ok] - c[ok]).abs().max()):.3e}")
