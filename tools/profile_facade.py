"""cProfile of single facade calls at full size (host-side overheads around the kernels)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import velocyto_amd as vcy
from velocyto_amd import ops
import bench
C, G = int(os.environ.get("C", 50000)), int(os.environ.get("G", 30000))
dev = ops.require_gpu()
S, U, pcs = bench.synth(C, G, 30, dev)
vlm = vcy.analysis.VelocytoLoom.from_arrays(S, U)
vlm.pcs = pcs.cpu().numpy(); vlm.ts = vlm.pcs[:, :2].copy()
vlm.normalize("both")
which = os.environ.get("WHICH", "knn_imputation")
calls = {"knn_imputation": lambda: vlm.knn_imputation(k=30, n_pca_dims=30),
         "balanced": lambda: vlm.knn_imputation(k=30, n_pca_dims=30, balanced=True, b_sight=240, b_maxl=120),
         "shift": lambda: (vlm.calculate_shift(), vlm.extrapolate_cell_at_t()),
         "embedding_shift": lambda: vlm.calculate_embedding_shift(),
         "prepare_markov": lambda: vlm.prepare_markov(2.0, 4.0),
         "etp": lambda: vlm.estimate_transition_prob(hidim="Sx_sz", embed="ts", n_neighbors=500, sampled_fraction=0.5)}
if which in ("shift", "embedding_shift", "prepare_markov", "etp"):
    vlm.knn_imputation(k=30, n_pca_dims=30); vlm.fit_gammas(fit_offset=False, weighted=False); vlm.predict_U(); vlm.calculate_velocity()
if which in ("embedding_shift", "prepare_markov", "etp"):
    vlm.calculate_shift(); vlm.extrapolate_cell_at_t()
if which in ("embedding_shift", "prepare_markov"):
    vlm.estimate_transition_prob(hidim="Sx_sz", embed="ts", n_neighbors=500, sampled_fraction=0.5, device_sampling=True)
if which == "prepare_markov":
    vlm.calculate_embedding_shift()
calls[which]()                      # warm-up (allocator, first launches)
torch.cuda.synchronize()
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
calls[which](); torch.cuda.synchronize()
pr.disable(); print(which, "wall", time.perf_counter() - t0)
st = pstats.Stats(pr).sort_stats("cumulative"); st.print_stats(22)
if os.environ.get("CALLERS"):
    st.print_callers(os.environ["CALLERS"])
