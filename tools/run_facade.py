"""Time the VelocytoLoom facade method by method on a synthetic dataset (default: BASELINE cfg2, 10k x 20k)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import velocyto_amd as vcy
from velocyto_amd import ops
import bench

C, G = int(os.environ.get("C", 10000)), int(os.environ.get("G", 20000))
dev = ops.require_gpu()
DT = {"f32": torch.float32, "f64": torch.float64}[os.environ.get("DTYPE", "f32")]
t_all = time.perf_counter()

def timed(name, fn, *a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = fn(*a, **k)
    torch.cuda.synchronize()
    mem = f"   [{torch.cuda.memory_allocated() / 2**30:6.1f} GiB in use, {torch.cuda.memory_reserved() / 2**30:6.1f} reserved]" if os.environ.get("MEM") else ""
    print(f"{name:32s} {1e3*(time.perf_counter()-t0):10.1f} ms{mem}", flush=True)
    return r

if os.environ.get("PRE", "1") == "1":
    # upstream callers on the raw uint16 count layers (tutorial order): detection filter, CV-vs-mean, size normalisation, PCA
    cS, cU, fS, fU, _ = bench.synth_counts(C, G, 30, dev)
    pre = vcy.analysis.VelocytoLoom.from_arrays(cS, cU)
    del cS, cU
    timed("score_detection_levels", pre.score_detection_levels, min_expr_counts=40, min_cells_express=30)
    timed("filter_genes(detection)", pre.filter_genes, by_detection_levels=True)
    print(f"    genes kept: {pre.dev('S').G} of {G}")
    timed("score_cv_vs_mean(N=3000)", pre.score_cv_vs_mean, N=3000, max_expr_avg=35)
    timed("filter_genes(cv_vs_mean)", pre.filter_genes, by_cv_vs_mean=True)
    timed("normalize_by_total", pre.normalize_by_total)
    timed("adjust_totS_totU", pre.adjust_totS_totU, normalize_total=True)
    timed(f"perform_PCA ({pre.dev('S').G} genes, all comps)", pre.perform_PCA)
    timed("perform_PCA(n_components=30)", pre.perform_PCA, n_components=30)
    del pre
    torch.cuda.empty_cache()

# the loom's count layers go in as they are on disk (uint16 / uint8 device matrices): knn_imputation then pools from the counts
# (vcy_knn_pool_counts) as long as S_sz / U_sz are still factor x counts; COUNTS=0 hands over f32 matrices instead (vcy_knn_pool2)
if os.environ.get("COUNTS", "1") == "1":
    cS, cU, fS, fU, pcs = bench.synth_counts(C, G, 30, dev)
    vlm = vcy.analysis.VelocytoLoom.from_arrays(cS, cU, dtype=DT)
    del cS, cU
else:
    S, U, pcs = bench.synth(C, G, 30, dev)
    vlm = vcy.analysis.VelocytoLoom.from_arrays(S, U, dtype=DT)
    del S, U
print(f"# facade storage {os.environ.get('DTYPE', 'f32')}, layers as {'counts' if os.environ.get('COUNTS', '1') == '1' else 'float matrices'}")
vlm.pcs = pcs.cpu().numpy(); vlm.ts = vlm.pcs[:, :2].copy()
for _pass in range(int(os.environ.get("PASSES", 1))):          # PASSES=2: the second pass is the steady state (buffers come from the allocator's cache)
    if _pass:
        print(f"-- pass {_pass + 1}")
        t_all = time.perf_counter()
    timed("normalize", vlm.normalize, "both")
    timed("knn_imputation(k=30)", vlm.knn_imputation, k=30, n_pca_dims=30)
    timed("knn_imputation(balanced)", vlm.knn_imputation, k=30, n_pca_dims=30, balanced=True, b_sight=240, b_maxl=120)
    timed("fit_gammas(default)", vlm.fit_gammas)
    timed("fit_gammas(plain)", vlm.fit_gammas, fit_offset=False, weighted=False)
    timed("predict_U", vlm.predict_U)
    timed("calculate_velocity", vlm.calculate_velocity)
    timed("calculate_shift", vlm.calculate_shift)
    timed("extrapolate_cell_at_t", vlm.extrapolate_cell_at_t)
    timed("estimate_transition_prob", vlm.estimate_transition_prob, hidim="Sx_sz", embed="ts", n_neighbors=500, sampled_fraction=0.5)
    timed("calculate_embedding_shift", vlm.calculate_embedding_shift)
    timed("prepare_markov", vlm.prepare_markov, 2.0, 4.0)
    timed(f"run_markov({int(os.environ.get('MARKOV_STEPS', 2500))})", vlm.run_markov, n_steps=int(os.environ.get("MARKOV_STEPS", 2500)))
    print("total", time.perf_counter() - t_all, "s;  delta_embedding[:2] =", vlm.delta_embedding[:2])
