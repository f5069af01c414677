"""Where the time of the facade's stage D/E methods goes: every velocyto_amd.ops function called from analysis.py is wrapped with a
device-synchronised timer (so the sums are wall time of serialised calls; the remainder is host Python / NumPy).
usage (GPU box): [C=50000 G=30000] python tools/facade_breakdown.py"""
import os, sys, time, collections, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import velocyto_amd as vcy
from velocyto_amd import ops, analysis
import bench

C, G = int(os.environ.get("C", 50000)), int(os.environ.get("G", 30000))
dev = ops.require_gpu()
DT = {"f32": torch.float32, "f64": torch.float64}[os.environ.get("DTYPE", "f32")]
cS, cU, fS, fU, pcs = bench.synth_counts(C, G, 30, dev)         # the loom's count layers: the facade pools from them (vcy_knn_pool_counts)
acc = collections.OrderedDict()


def wrap(mod, name):
    fn = getattr(mod, name)

    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize()
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return r
    setattr(mod, name, w)


for n, f in list(vars(ops).items()):
    if isinstance(f, types.FunctionType) and not n.startswith("_") and n not in ("require_gpu", "padded_ld"):
        wrap(ops, n)
for n in ("_permute_rows_nsign", "_fill_diagonal_zero"):
    if hasattr(analysis, n):
        wrap(analysis, n)

vlm = vcy.analysis.VelocytoLoom.from_arrays(cS, cU, dtype=DT)
del cS, cU
vlm.pcs = pcs.cpu().numpy(); vlm.ts = vlm.pcs[:, :2].copy()


def run(name, fn, *a, **k):
    acc.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fn(*a, **k)
    torch.cuda.synchronize(); tot = time.perf_counter() - t0
    print(f"{name}: {1e3 * tot:.1f} ms")
    for n, t in sorted(acc.items(), key=lambda x: -x[1]):
        print(f"    {n:36s} {1e3 * t:9.1f} ms")
    print(f"    {'(host remainder)':36s} {1e3 * (tot - sum(acc.values())):9.1f} ms")


for rep in range(2):
    run("normalize", vlm.normalize, "both")
    run("knn_imputation", vlm.knn_imputation, k=30, n_pca_dims=30)
    run("fit_gammas(default)", vlm.fit_gammas)
    run("fit_gammas(plain)", vlm.fit_gammas, fit_offset=False, weighted=False)
    run("predict_U..extrapolate", lambda: (vlm.predict_U(), vlm.calculate_velocity(), vlm.calculate_shift(), vlm.extrapolate_cell_at_t()))
    run("estimate_transition_prob", vlm.estimate_transition_prob, hidim="Sx_sz", embed="ts", n_neighbors=500, sampled_fraction=0.5)
    run("calculate_embedding_shift", vlm.calculate_embedding_shift)
    run("prepare_markov", vlm.prepare_markov, 2.0, 4.0)
    run("run_markov", vlm.run_markov)
