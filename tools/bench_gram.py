"""csrc/gram.hip alone: the centred Gram product of perform_PCA (vcy_gram, f64 MFMA) and the thin block product of the subspace
iteration (vcy_gram_tn) against the library route they replace (torch.addmm_ on explicitly centred blocks = rocBLAS dgemm), with
flop rates against the f64 matrix peak of the chip (78.6 Tflop/s = 1024 SIMDs x 2048 flop / 64 clocks x 2.4 GHz).
SHAPES="C,G;C,G" picks the shapes; ONLY=gram|tn restricts (for rocprofv3 --pmc passes: tools/pmc_kernel.sh k_gram '...')."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import velocyto_amd
from velocyto_amd import ops

dev = ops.require_gpu()
PEAK = 78.6e12
shapes = [tuple(int(v) for v in s.split(",")) for s in os.environ.get("SHAPES", "50000,3000;50000,10000;50000,30000;10000,20000").split(";")]
only = os.environ.get("ONLY", "")
reps = int(os.environ.get("REPS", 3))


def best(f, n=reps):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, r


def library_gram(M, mean, block=8192):
    G = M.G
    g = torch.zeros((G, G), dtype=torch.float64, device=dev)
    for s in range(0, M.C, block):
        A = M.t[s:s + block, :G].double() - mean
        g.addmm_(A.T, A)
    return g


for C, G in shapes:
    for dt in (torch.float64, torch.float32):
        gen = torch.Generator(device=dev).manual_seed(1)
        X = ops.CellMatrix.empty(C, G, dt)
        for s in range(0, C, 8192):
            X.t[s:s + 8192] = torch.randn(X.t[s:s + 8192].shape, generator=gen, device=dev, dtype=torch.float32).to(dt) + 3.0
        X.t[:, G:] = 0
        mean = ops.col_means(X)
        name = "f64" if dt == torch.float64 else "f32"
        if only in ("", "gram"):
            ms, g = best(lambda: ops.gram(X, mean))
            nt = (G + 127) // 128
            fl_exec, fl_full = 2.0 * C * (nt * (nt + 1) // 2) * 128 * 128, 2.0 * C * G * G       # flops on the upper-triangle tiles; of the full product
            line = f"gram   C {C:6d} G {G:6d} {name}: {ms:9.2f} ms  {fl_exec / ms / 1e9:7.1f} Tflop/s executed = {fl_exec / (ms * 1e-3) / PEAK:5.3f} of the f64 matrix peak; " \
                   f"as a full product {fl_full / ms / 1e9:7.1f} Tflop/s"
            if G <= 12000 and not only:
                ms_l, gl = best(lambda: library_gram(X, mean), 2)
                err = float((g - gl).abs().max() / gl.diagonal().max())
                line += f" | library (addmm_ of centred f64 blocks) {ms_l:9.2f} ms = {ms_l / ms:4.2f} x, max |diff| / max diag {err:.1e}"
                del gl
            print(line, flush=True)
            del g
        if only in ("", "tn"):
            L = 50
            Z = torch.linalg.qr(torch.randn((G, L), generator=gen, device=dev, dtype=torch.float64))[0]
            Y = torch.empty((C, L), dtype=torch.float64, device=dev)
            mz = mean @ Z
            for s in range(0, C, 8192):
                Y[s:s + 8192] = X.t[s:s + 8192, :G].double() @ Z - mz
            ms, w = best(lambda: ops.gram_tn(X, mean, Y))
            byts = C * G * X.t.element_size()
            line = f"gram_tn C {C:6d} G {G:6d} L {L} {name}: {ms:8.2f} ms  {2.0 * C * G * L / ms / 1e9:6.1f} Tflop/s useful, X streamed at {byts / ms / 1e6:7.1f} GB/s"
            if not only:
                def lib():
                    out = torch.zeros((G, L), dtype=torch.float64, device=dev)
                    for s in range(0, C, 8192):
                        b = X.t[s:s + 8192, :G].double()
                        out.addmm_(b.T, Y[s:s + 8192])
                    return out - torch.outer(mean, Y.sum(0))
                ms_l, wl = best(lib, 2)
                line += f" | library {ms_l:8.2f} ms = {ms_l / ms:4.2f} x, max |diff| {float((w - wl).abs().max() / wl.abs().max()):.1e}"
            print(line, flush=True)
        del X
        torch.cuda.empty_cache()
