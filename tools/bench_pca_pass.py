"""One pass of perform_PCA's subspace iteration at the headline size (50 000 cells x 30 000 genes, block of k + 20 = 50 columns):
the projection over the genes Y = (X - m) Z as the hand-written kernel (vcy_gemm_nt: X read as stored) against the route it replaced
(fp64 copies of 8192-cell blocks through the library GEMM), the contraction over the cells (vcy_gram_tn), the final scores, and the dual
route's Gram matrix of the cells at 3 000 x 30 000."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import velocyto_amd
from velocyto_amd import ops
dev = ops.require_gpu()
C, G, L = int(os.environ.get("C", 50000)), int(os.environ.get("G", 30000)), int(os.environ.get("L", 50))


def best(f, reps=3):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3


for dt in (torch.float64, torch.float32):
    s = 8 if dt == torch.float64 else 4
    X = ops.CellMatrix.empty(C, G, dt)
    X.t[:, :G] = torch.rand((C, G), device=dev, dtype=dt) * 3 + 1
    mean = ops.col_means(X)
    Z = torch.linalg.qr(torch.randn((G, L), device=dev, dtype=torch.float64))[0]
    Y = torch.empty((C, L), dtype=torch.float64, device=dev)

    def old_route(block=8192):
        mz = mean @ Z
        for b in range(0, C, block):
            Y[b:b + block] = X.t[b:b + block, :G].double() @ Z - mz
    t_old = best(old_route)
    y_old = Y.clone()
    t_new = best(lambda: ops.gemm_nt(X, ops.CellMatrix.from_genes_major(Z, torch.float64), col_corr=(Z * mean[:, None]).sum(0), out=Y))
    err = float((Y - y_old).abs().max() / y_old.abs().max())
    t_tn = best(lambda: ops.gram_tn(X, mean, Y))
    byt = C * G * s
    print(f"{str(dt):14s} C={C} G={G} L={L}:  Y = (X - m) Z  library route {t_old:7.2f} ms ({byt / t_old / 1e9:5.2f} TB/s of X as stored)   vcy_gemm_nt {t_new:7.2f} ms "
          f"({byt / t_new / 1e9:5.2f} TB/s = {byt / t_new / 1e9 / 8.0:.2f} of HBM peak; {2.0 * C * G * L / t_new / 1e9:.1f} Tflop/s)   max rel diff {err:.1e}   "
          f"A^T Y (vcy_gram_tn) {t_tn:7.2f} ms   -> one pass {t_old + t_tn:7.2f} -> {t_new + t_tn:7.2f} ms")
    del X, Y, y_old
    torch.cuda.empty_cache()
Cw = 3000
Xw = ops.CellMatrix.empty(Cw, G, torch.float64)
Xw.t[:, :G] = torch.rand((Cw, G), device=dev, dtype=torch.float64) * 3 + 1
mean = ops.col_means(Xw)


def old_dual():
    A = Xw.t[:, :G].double() - mean
    return A @ A.T


def new_dual():
    a = ops.gemm_nt(Xw, mean[None, :])[:, 0].contiguous()
    return ops.gemm_nt(Xw, Xw, row_corr=a, col_corr=a, c0=float(mean @ mean))
t_o, t_n = best(old_dual), best(new_dual)
d = float((old_dual() - new_dual()).abs().max() / old_dual().abs().max())
print(f"dual route, (X - m)(X - m)^T at {Cw} cells x {G} genes (f64): centred copy + library GEMM {t_o:7.2f} ms   vcy_gemm_nt x 2 {t_n:7.2f} ms "
      f"({2.0 * Cw * Cw * G / t_n / 1e9:.1f} Tflop/s = {2.0 * Cw * Cw * G / t_n / 1e9 / 78.6:.2f} of the f64 matrix peak)   max rel diff {d:.1e}")
