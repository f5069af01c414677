"""Stage E's expression scaling alone at the headline size: vcy_embedding_scaling (one launch, estimates in registers) against the
two-step route it replaces (vcy_knn_pool_w2 + 2 x vcy_row_cosproj).  DTYPE=f32|f64, SINGLE=1 for one weight set."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import velocyto_amd
from velocyto_amd import ops
import bench

dev = ops.require_gpu()
DT = torch.float64 if os.environ.get("DTYPE", "f32") == "f64" else torch.float32
C, G = int(os.environ.get("C", 50000)), int(os.environ.get("G", 30000))
S, U, pcs = bench.synth(C, G, 30, dev)
if DT == torch.float64:
    S = ops.CellMatrix(S.t.double(), G)
del U
emb = pcs[:, :2].contiguous()
neigh, _ = bench.sample_neighbors_device(emb, 500, 0.5, dev)
n = neigh.shape[1]
gen = torch.Generator(device=dev).manual_seed(3)
dS = ops.CellMatrix(torch.randn(S.t.shape, generator=gen, device=dev).to(DT), G)
dR = ops.CellMatrix(torch.randn(S.t.shape, generator=gen, device=dev).to(DT), G)
w = (torch.rand((C, n), generator=gen, device=dev) / n - 0.5 / n).to(DT)
w2 = (torch.rand((C, n), generator=gen, device=dev) / n - 0.5 / n).to(DT)
order = ops.hilbert_order(emb)
single = bool(os.environ.get("SINGLE"))


def best(f, reps=3):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, r


ms_new, got = best(lambda: ops.embedding_scaling(S, dS, neigh, w, None if single else dR, None if single else w2, order=order))
flops = 2.0 * C * n * G * (1 if single else 2)
print(f"{os.environ.get('DTYPE', 'f32')} {'single' if single else 'dual'}: embedding_scaling {ms_new:8.2f} ms = {flops / ms_new / 1e9:6.1f} Tflop/s of multiply-adds", flush=True)
if not os.environ.get("ONLY_NEW"):
    indptr = torch.arange(0, (C + 1) * n, n, dtype=torch.int64, device=dev)
    def old():
        if single:
            e1 = ops.knn_pool(S, indptr, neigh.reshape(-1), w.reshape(-1), validate=False, order=order)
            return (ops.row_cosproj(dS, e1),)
        e1, e2 = ops.knn_pool_w2(S, indptr, neigh.reshape(-1), w.reshape(-1), w2.reshape(-1), validate=False, order=order)
        return ops.row_cosproj(dS, e1), ops.row_cosproj(dR, e2)
    ms_old, ref = best(old, 2)
    err = max(float((a - b).abs().max()) for a, b in zip(got, ref))
    print(f"    two-step route {ms_old:8.2f} ms = {ms_old / ms_new:4.2f} x; max |cos_new - cos_old| = {err:.2e}")
