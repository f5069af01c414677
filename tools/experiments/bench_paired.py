#!/usr/bin/env python3
"""Stage D at the headline size (50 000 x 30 000, nrndm 250), three ways on the same pooled matrices:
  fused     vcy_coldeltacor_partial_fused (round 4's headline launch: velocity chain folded in, every listed pair evaluated)
  plain     vcy_velocity_chain (dmat materialised) + vcy_coldeltacor_partial
  paired    vcy_velocity_chain + vcy_cell_moments + vcy_coldeltacor_pair_plan + vcy_coldeltacor_partial_paired (mirrored pairs once)
and how far the three results are from one another.  DTYPE=f64|f32, REPS=n, CELLS/GENES to shrink."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench

a = types.SimpleNamespace(cells=int(os.environ.get("CELLS", 50000)), genes=int(os.environ.get("GENES", 30000)), k=30, pca_dims=30, n_neighbors=500,
                          sampled_fraction=0.5, curve="hilbert", order="embedding", exchange="halo", overlap=True, slab=0, fuse=True, literal_rule=True,
                          counts="auto")
dev = torch.device("cuda", 0)
dtype = torch.float64 if os.environ.get("DTYPE", "f64") == "f64" else torch.float32
reps = int(os.environ.get("REPS", 5))
pipe = bench.Pipeline(a, dev, 0, 1, dtype=dtype)
ops = pipe.ops
gamma = pipe.step()
rules = pipe.rules
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn, n=reps):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0.record(); r = fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return r, float(np.median(ts)), float(np.min(ts))


def fused():
    return ops.coldeltacor_partial_fused(pipe.e_rows, pipe.Ux_loc, gamma, None, pipe.neigh_k, ops.SQRT, rules, 1e-10, order=pipe.order, out=pipe.corr_loc, validate=False)


r_f, t_f, m_f = timed(fused)
r_f = r_f.clone()
print(f"fused: {t_f:.2f} ms (min {m_f:.2f})", flush=True)
dmat = ops.velocity_chain(pipe.Sx_loc, pipe.Ux_loc, gamma, None, want=("dmat",), transform=ops.SQRT, psc=1e-10)["dmat"]
_, t_c, _ = timed(lambda: ops.velocity_chain(pipe.Sx_loc, pipe.Ux_loc, gamma, None, want=("dmat",), transform=ops.SQRT, psc=1e-10))
print(f"velocity_chain -> dmat: {t_c:.2f} ms", flush=True)
out2 = torch.empty_like(pipe.corr_loc)
r_p, t_p, m_p = timed(lambda: ops.coldeltacor_partial(pipe.e_rows, dmat, pipe.neigh_k, ops.SQRT, rules, 1e-10, order=pipe.order, out=out2, validate=False))
r_p = r_p.clone()
print(f"plain (materialised d): {t_p:.2f} ms (min {m_p:.2f})", flush=True)
_, t_m, _ = timed(lambda: ops.cell_moments(dmat))
plan, t_pl, _ = timed(lambda: ops.pair_plan(pipe.neigh_k, 0))
dm = ops.cell_moments(dmat)
print(f"cell_moments: {t_m:.2f} ms; pair_plan: {t_pl:.2f} ms; plan: {float((plan >= 0).float().mean()):.4f} evaluated with their mirror, "
      f"{float((plan == -2).float().mean()):.4f} handed over, {float((plan == -1).float().mean()):.4f} ordinary", flush=True)
out3 = torch.empty_like(pipe.corr_loc)
r_q, t_q, m_q = timed(lambda: ops.coldeltacor_partial_paired(pipe.e_rows, dmat, pipe.neigh_k, ops.SQRT, rules, 1e-10, order=pipe.order, out=out3, validate=False, plan=plan, dm=dm))
print(f"paired: {t_q:.2f} ms (min {m_q:.2f})", flush=True)
nothing = torch.full_like(plan, -1)
r_n, t_n, m_n = timed(lambda: ops.coldeltacor_partial_paired(pipe.e_rows, dmat, pipe.neigh_k, ops.SQRT, rules, 1e-10, order=pipe.order, out=torch.empty_like(out3), validate=False, plan=nothing, dm=dm))
print(f"paired kernel with a plan that pairs nothing: {t_n:.2f} ms (min {m_n:.2f})", flush=True)
ok = torch.isfinite(r_f)
print("nan pattern equal:", bool(torch.equal(torch.isnan(r_f), torch.isnan(r_q))), bool(torch.equal(torch.isnan(r_f), torch.isnan(r_p))))
print(f"max |paired - fused| = {float((r_q[ok] - r_f[ok]).abs().max()):.3e}; max |plain - fused| = {float((r_p[ok] - r_f[ok]).abs().max()):.3e}; "
      f"paired == paired-with-empty-plan bit for bit: {bool(torch.equal(r_q[ok], r_n[ok]))}")
print(f"stage C + D: fused {t_f:.2f} ms; paired {t_c + t_m + t_pl + t_q:.2f} ms (chain {t_c:.2f} + moments {t_m:.2f} + plan {t_pl:.2f} + launch {t_q:.2f})")
