"""Time the all-pairs (full) correlation kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import velocyto_amd
from velocyto_amd import ops
dev = ops.require_gpu()
C, G = int(os.environ.get("C", 10000)), int(os.environ.get("G", 20000))
e = ops.CellMatrix(torch.rand((C, ops.padded_ld(G)), device=dev), G)
d = ops.CellMatrix(torch.randn((C, ops.padded_ld(G)), device=dev), G)
for tr, name in ((ops.SQRT, "sqrt"), (ops.LINEAR, "linear"), (ops.LOG10, "log10")):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rm = ops.coldeltacor_full(e, d, tr, 1e-10)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"full {name:7s} C={C} G={G}: {dt*1e3:8.1f} ms  {C*C*G/dt/1e12:.2f} T pair-genes/s")
ops.FULL_LINEAR_GEMM = False
torch.cuda.synchronize(); t0 = time.perf_counter(); rm = ops.coldeltacor_full(e, d, ops.LINEAR); torch.cuda.synchronize()
print(f"full linear (VALU kernel) C={C} G={G}: {(time.perf_counter()-t0)*1e3:8.1f} ms")
