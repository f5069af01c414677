#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X: cells/sec through knn_imputation -> fit_slope -> colDeltaCor.

Workload (BASELINE.json configs[2], SURVEY.md section 8d "cfg3"): synthetic 50 000 cells x 30 000 genes,
k = 30 kNN in a 30-d PCA space, estimate_transition_prob(transform="sqrt", n_neighbors=500,
sampled_fraction=0.5) => nrndm = 250.  One timed "step" = one pass of the path over the whole
dataset, inputs (count layers + size factors, pcs, sampled embedding neighbours) already resident in HBM:

  A  knn_imputation : exact kNN search in pcs + connectivity weights + pooling of S_sz and U_sz (gathered from the
                      resident count layers and per-cell size factors: S_sz = factor * counts)
  B  fit_slope      : per-gene gamma = max(0, <Sx,Ux>/<Sx,Sx>)
  C  velocity chain : predict_U -> velocity -> delta_S -> signed-sqrt dmat (one fused pass; by default folded into D's staging)
  D  colDeltaCorSqrtpartial on the sampled embedding neighbours (the dominant kernel)

`--workload cfg5` runs the atlas-scale form of the same path (BASELINE.json configs[4]): CSR count layers at ~8 % density,
streamed over Hilbert-ordered cell blocks with sharded `e` (velocyto_amd/atlas.py).

N > 1 (one rank per GPU over RCCL; launched by torch.distributed.run, or self-launched with torch.multiprocessing when
WORLD_SIZE is not set): cells are sharded, total work fixed ("strong" scaling): ranks all-reduce the fit moments, exchange
the halo rows of Sx their neighbour lists reference and all-gather the compact correlation rows.

Prints ONE JSON line (rank 0).  The headline (`value`, `dtype`, `ms_per_step`, `roofline`) is the pass in the REFERENCE'S
arithmetic: f64 storage of the pooled matrices, f64 moments, the literal branch rule of speedboosted.pyx:372-378
(`--dtype f64`, the default), timed with exactly --steps / --warmup.  `roofline` is for the dominant kernel
(k_cdc_partial_grouped<double>): it is VALU-issue-bound, so `achieved` is its VALU wave-instruction rate (instructions per
launch from the rocprofv3 SQ_INSTS_VALU pass named in `roofline.counters_from`, scaled by the exact pair-chunk count of this
run; time from HIP events of this run) against the issue peak of the chip; the HBM-side figures ride along
(`hbm_frac_measured`, `vs_noreuse_model`).  `precision_modes` holds the build's narrower production modes of the same pass
(f32 storage with the no-pseudocount form of the rule, f32 with the literal rule, uint16 count layers), each with its own
roofline block and its distance from the f64 pass over ALL correlations - reported, not the headline.  `stages` holds the
per-stage rooflines.  `extra` holds the other lines SURVEY.md 8(d) asks for, all timed in this run at the headline size:
E calculate_embedding_shift, F prepare_markov + run_markov, B fit_gammas with its defaults, D at the reference's default list
width (n_neighbors = C/5, sampled_fraction = 0.3 => nrndm = 3000), the randomised control, and cfg2 (BASELINE.json
configs[1]: 10 000 x 20 000, stages A + B, unbalanced and balanced kNN); their key numbers are repeated as scalars in `config`.
`cpu_baseline` times the CPU side on a bounded closed sub-problem on the host cores: stage D - 98 % of it - by the reference's
own Cython kernel where oracle/_ref travelled with the snapshot (`kind: "reference"`; built from /root/reference in the build
container by oracle/build_ref.py), else by the oracle restatement (`kind: "port"`, oracle/libvelocyto_oracle.so + oracle.py);
`parity` says how far the HIP path is from the restatement on that same sub-problem in all three modes.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK = 8.0e12                 # B/s, MI355X spec (MI355X_MICROARCH.md)
# VALU issue peak: 256 CUs x 4 SIMD-32, one wave64 instruction per 2 clocks per SIMD at 2.4 GHz (MI355X_MICROARCH.md
# "Per-instruction cycle constants": v_fma_f32 2 cyc; tools/ubench/valu_issue.hip measures 2.25 for the plain 2-operand ops)
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 2.0
# Issue cost of the instruction mix the transform + three moment updates NEED per element, measured as a mix by
# tools/ubench/valu_issue.hip (profiles/r02_valu_issue.txt; clocks per wave64 instruction per SIMD, 4 waves per SIMD), by branch rule:
#   1 literal partial sqrt (speedboosted.pyx:372-378): sub, |t|*2^54 clamp, fma with psc, v_sqrt_f32, v_bfi_b32, add, two fma -
#     "cdc element pair" 4.59 clocks x 6.04 instructions
#   2 pseudocount dropped (VCY_RULES_PARTIAL_NOPSC, f32): v_sub, v_rsq_f32, v_mul_legacy_f32, v_add, two v_fmac -
#     "cdc no-psc element" 3.89 clocks x 6 instructions (the parts alone sum to 19.8: a transcendental between plain ops costs more)
MIX_CLK_PER_ELEMENT = {1: 27.7, 2: 23.3}
# f64 (the reference's arithmetic): literal element = v_add_f64 (sub), v_add_f64 (|t| + psc), v_cvt_f32_f64, v_cmp_f64 + v_cndmask (zero rule, on the argument of the
# seed), v_rsq_f32, 2 x v_mul_f32 (the 24-bit seed s0 and h = y / 2), 2 x v_cvt_f64_f32, 4 x v_fma_f64 (two Newton corrections), v_bfi (sign), v_add_f64 + 2 x v_fma_f64
# (moments) - 18 instructions: measured as a mix by tools/ubench/valu_issue_f64.hip (profiles/r04_valu_issue_f64.txt: "f64 element, zero rule on the seed's argument",
# wall-clock column; the 19-instruction form it replaced: 83.2)
MIX_CLK_PER_ELEMENT_F64 = 77.4
F64_ISSUE_CLK = 4.1               # clocks per wave64 v_add_f64 / v_mul_f64 / v_fma_f64 per SIMD (same file): the f64 issue peak is 1 instruction / 4 clocks
COUNTERS_FILE = os.path.join(ROOT, "profiles", "r06_cdc_counters.json")     # written by tools/summarize_profiles.py from the rocprofv3 passes
COUNTERS_FALLBACK = os.path.join(ROOT, "profiles", "r05_cdc_counters.json")
WIDE_COUNTERS_FILE = os.path.join(ROOT, "profiles", "r06_cdc_wide_counters.json")   # tools/pmc_wide.sh: the launch at the reference's default list width


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cells", type=int, default=50000)
    ap.add_argument("--genes", type=int, default=30000)
    ap.add_argument("--k", type=int, default=30)
    ap.add_argument("--pca-dims", type=int, default=30)
    ap.add_argument("--n-neighbors", type=int, default=500)
    ap.add_argument("--sampled-fraction", type=float, default=0.5)
    ap.add_argument("--cpu-cells", type=int, default=256, help="cells of the closed CPU-baseline sub-problem (stages A-C, parity of the HIP path); "
                                                                "stage D is timed at full width beside it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per launch of the dominant kernel from a separate rocprofv3 --pmc pass of THIS command; without it "
                         "the figure recorded for the default workload in profiles/r04_cdc_counters.json is reported (named in "
                         "roofline.counters_from), null for any other workload")
    ap.add_argument("--workload", choices=["cfg3", "cfg5"], default="cfg3",
                    help="cfg3: dense count layers, everything resident (the headline); cfg5: CSR layers, block-streamed (atlas.py)")
    ap.add_argument("--density", type=float, default=0.08, help="cfg5: fraction of non-zero counts per cell")
    ap.add_argument("--block-cells", type=int, default=0, help="cfg5: cells per streamed block (0 = chosen from free HBM)")
    ap.add_argument("--knn", choices=["auto", "brute", "pruned"], default="auto", help="cfg5: exact kNN search by brute force or projection-pruned (auto: pruned from 100k cells)")
    ap.add_argument("--counts", choices=["auto", "u16"], default="u16",
                    help="storage of the resident count layers: u16 = uint16, the loom's on-disk type (constants.py:11; the headline); auto = what "
                         "ops.CountMatrix.narrowed() keeps for this dataset: uint8 when no count exceeds 255 (lossless), else uint16")
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f64", help="arithmetic / storage type of the timed path (f64 = the reference's, the headline; "
                                                                         "f32 = the build's production mode)")
    ap.add_argument("--no-extra", action="store_true", help="skip precision_modes and the extra lines (E, F, default fit_gammas, nrndm = 3000, randomised control, cfg2)")
    ap.add_argument("--extra-budget-s", type=float, default=40.0, help="extra lines are skipped (and say so) once this many seconds have gone into them")
    ap.add_argument("--literal-rule", action="store_true", help="stage D with the literal partial-sqrt rule (pseudocount kept) instead of the "
                                                                "three-instruction f32 form ops.partial_rules_for picks")
    ap.add_argument("--slab", type=int, default=0, help="gene slab of the pooling kernel (0 = library default)")
    ap.add_argument("--no-fuse", dest="fuse", action="store_false",
                    help="materialise dmat with k_velocity_chain instead of folding the velocity chain into stage D")
    ap.add_argument("--exchange", choices=["halo", "allgather"], default="halo",
                    help="N > 1: how ranks obtain the rows of e = Sx_sz their neighbour lists reference")
    ap.add_argument("--no-overlap", dest="overlap", action="store_false",
                    help="N > 1 with the halo exchange: do not split stage D into interior cells (run while the halo moves) and the rest")
    ap.add_argument("--dump", default=None, help="rank 0 saves gamma and the gathered correlation rows of the last step to this .npz (tests)")
    ap.add_argument("--generator", choices=["survey", "bench"], default="survey",
                    help="synthetic dataset: 'survey' (default from round 6) = SURVEY.md 8(d)'s - numpy PCG64(20180808 + cfg#), 12 clusters on a branching latent "
                         "time, the splicing ODE's closed form; 'bench' = the device-side generator rounds 1-5 were timed on (one branch point, seed 20180811)")
    ap.add_argument("--curve", choices=["morton", "hilbert"], default="hilbert",
                    help="space-filling curve of the stage-D schedule and of the cell relabelling of sharded runs (Hilbert: no jumps, "
                         "8-cell groups share more neighbours: 97.3 vs 98.6 ms)")
    ap.add_argument("--order", choices=["natural", "embedding"], default="embedding",
                    help="schedule order of the cells in stage D (results are order-independent)")
    return ap.parse_args()


def synth_counts(C, G, P, dev, seed=20180811):
    """Seeded synthetic loom-like dataset generated on the device in cell blocks: uint16 spliced/unspliced COUNT
    matrices (cells-major), the per-cell size factors of the a1 pre-step (analysis.py:535-582; not in the metric)
    and `pcs`.  Returns (cS, cU, fS, fU, pcs)."""
    from velocyto_amd import ops
    gen = torch.Generator(device=dev).manual_seed(seed)
    ld = ops.padded_ld(G)
    S = torch.zeros((C, ld), dtype=torch.int16, device=dev)
    U = torch.zeros((C, ld), dtype=torch.int16, device=dev)
    alpha = torch.exp(torch.randn(G, generator=gen, device=dev))
    gamma = torch.exp(-0.5 + 0.5 * torch.randn(G, generator=gen, device=dev))
    t_on = torch.rand(G, generator=gen, device=dev) * 0.7
    switching = (torch.rand(G, generator=gen, device=dev) < 0.6).float()
    t = torch.rand(C, generator=gen, device=dev)
    branch = (torch.rand(C, generator=gen, device=dev) < 0.5).float()
    branch_gene = (torch.rand(G, generator=gen, device=dev) < 0.3).float()
    size = torch.exp(0.3 * torch.randn(C, generator=gen, device=dev))
    sumS = torch.zeros(C, dtype=torch.float64, device=dev)
    sumU = torch.zeros(C, dtype=torch.float64, device=dev)
    blk = 4096
    for s in range(0, C, blk):
        tt = t[s:s + blk, None]
        tau = torch.clamp(tt - t_on[None, :], min=0.0) * switching[None, :] + (1 - switching[None, :]) * 1.0
        gate = 1.0 - branch_gene[None, :] * branch[s:s + blk, None] * (tt > 0.5).float()
        u = alpha[None, :] * (1 - torch.exp(-4.0 * tau)) * gate
        sp = (alpha / gamma)[None, :] * (1 - torch.exp(-2.0 * gamma[None, :] * tau)) * gate
        sz = size[s:s + blk, None]
        cu = torch.poisson(0.3 * sz * u, generator=gen).clamp_(max=65535)
        cs = torch.poisson(sz * sp, generator=gen).clamp_(max=65535)
        sumU[s:s + blk], sumS[s:s + blk] = cu.sum(1).double(), cs.sum(1).double()
        U[s:s + blk, :G] = cu.to(torch.int32).to(torch.int16)          # uint16 bit pattern
        S[s:s + blk, :G] = cs.to(torch.int32).to(torch.int16)
    fS = sumS.mean() / sumS.clamp(min=1.0)                              # avg_size / cell_size
    fU = sumU.mean() / sumU.clamp(min=1.0)
    # resident encoding of the layers: uint8 when no count exceeds 255 (lossless; true of this generator: max 189), else uint16
    cS, cU = ops.CountMatrix(S, G).narrowed(), ops.CountMatrix(U, G).narrowed()
    if cS.t.dtype != cU.t.dtype:
        cS, cU = ops.CountMatrix(S, G), ops.CountMatrix(U, G)
    # pcs: top-P principal components of log2(S_sz + 1) (perform_PCA is upstream of the path; randomised SVD here)
    L = torch.empty((C, G), dtype=torch.float32, device=dev)
    for s in range(0, C, blk):
        L[s:s + blk] = torch.log2(cS.as_int32(s, s + blk).float() * fS[s:s + blk, None].float() + 1.0)
    L -= L.mean(0, keepdim=True)
    torch.manual_seed(seed)             # svd_lowrank draws its sketch from the global RNG: keep every rank's pcs identical
    Uu, Ss, _ = torch.svd_lowrank(L, q=P, niter=2)
    pcs = (Uu * Ss).double().contiguous()
    del L
    return cS, cU, fS, fU, pcs


SURVEY_SEED0 = 20180808          # seed = 20180808 + cfg#

# the branching latent time of the survey generator: 12 clusters = 12 (lineage, time interval) pieces of a tree with three leaves.
# Lineage bits: 1 = A1, 2 = A2, 4 = B (a cell on the trunk belongs to all three, a cell on branch A to A1 and A2).
_SURVEY_CLUSTERS = ((7, 0.00, 0.12), (7, 0.12, 0.25),                                  # trunk
                    (3, 0.25, 0.37), (3, 0.37, 0.49), (3, 0.49, 0.60),                 # branch A before it splits
                    (1, 0.60, 0.80), (1, 0.80, 1.00), (2, 0.60, 0.80), (2, 0.80, 1.00),   # leaves A1, A2
                    (4, 0.25, 0.50), (4, 0.50, 0.75), (4, 0.75, 1.00))                 # branch B


def synth_counts_survey(C, G, P, dev, cfg=3):
    """SURVEY.md section 8(d)'s generator: numpy.random.Generator(PCG64(20180808 + cfg#)); 12 clusters on a branching latent time
    t_c in [0, 1]; per gene alpha ~ LogNormal(0, 1), beta = 1, gamma ~ LogNormal(-0.5, 0.5), 60 % of the genes switch on or off at a
    random t (the analytic u(t), s(t) of the splicing ODE du/dt = alpha - u, ds/dt = u - gamma s, ODE time = 10 x latent time), the
    rest constitutive (steady state); per-cell size factor ~ LogNormal(0, 0.3); U ~ Poisson(size u 0.3), S ~ Poisson(size s); uint16.
    Every PARAMETER (cluster, latent time, alpha, gamma, switch time, direction, lineage mask, size) is drawn on the host from that PCG64 stream,
    identically on every box; the 3e9 Poisson draws themselves are made on the device (torch's Philox generator, seeded from the same stream) -
    on the host they would take minutes.  Returns (cS, cU, fS, fU, pcs) like synth_counts."""
    from velocyto_amd import ops
    rng = np.random.Generator(np.random.PCG64(SURVEY_SEED0 + cfg))
    T_ODE = 10.0
    cl = rng.integers(0, len(_SURVEY_CLUSTERS), C)
    tab = np.array(_SURVEY_CLUSTERS, dtype=np.float64)
    t_np = tab[cl, 1] + rng.random(C) * (tab[cl, 2] - tab[cl, 1])
    lin_np = tab[cl, 0].astype(np.int64)
    alpha_np = np.exp(rng.normal(0.0, 1.0, G))
    gamma_np = np.exp(rng.normal(-0.5, 0.5, G))
    gamma_np = np.where(np.abs(gamma_np - 1.0) < 1e-3, 1.001, gamma_np)          # (the closed form divides by gamma - beta)
    switching_np = rng.random(G) < 0.6
    t_sw_np = rng.random(G) * 0.8
    turns_on_np = rng.random(G) < 0.5                                            # else: starts at steady state and is switched off
    mask_np = np.where(rng.random(G) < 0.5, 7, rng.integers(1, 7, G))            # lineages on which the switch happens (half of them: everywhere)
    size_np = np.exp(rng.normal(0.0, 0.3, C))
    gen = torch.Generator(device=dev).manual_seed(int(rng.integers(0, 2 ** 62)))
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    t, size = f32(t_np), f32(size_np)
    lin = torch.from_numpy(lin_np).to(dev)
    alpha, gam, t_sw = f32(alpha_np), f32(gamma_np), f32(t_sw_np)
    switching, turns_on = torch.from_numpy(switching_np).to(dev), torch.from_numpy(turns_on_np).to(dev)
    gmask = torch.from_numpy(mask_np.astype(np.int64)).to(dev)
    ld = ops.padded_ld(G)
    S = torch.zeros((C, ld), dtype=torch.int16, device=dev)
    U = torch.zeros((C, ld), dtype=torch.int16, device=dev)
    sumS = torch.zeros(C, dtype=torch.float64, device=dev)
    sumU = torch.zeros(C, dtype=torch.float64, device=dev)
    u_ss, s_ss = alpha[None, :], (alpha / gam)[None, :]
    blk = 4096
    for b in range(0, C, blk):
        tt = t[b:b + blk, None]
        # the switch has happened for this cell if its lineage carries it and its latent time is past the switch time
        active = switching[None, :] & ((lin[b:b + blk, None] & gmask[None, :]) != 0) & (tt > t_sw[None, :])
        tau = torch.clamp(tt - t_sw[None, :], min=0.0) * T_ODE
        e1, eg = torch.exp(-tau), torch.exp(-gam[None, :] * tau)
        u_on = alpha[None, :] * (1 - e1)
        s_on = (alpha / gam)[None, :] * (1 - eg) + (alpha / (gam - 1.0))[None, :] * (eg - e1)
        u_off = alpha[None, :] * e1
        s_off = (s_ss - u_ss / (gam[None, :] - 1.0)) * eg + u_ss * e1 / (gam[None, :] - 1.0)
        on = turns_on[None, :]
        u = torch.where(active, torch.where(on, u_on, u_off), torch.where(on & switching[None, :], torch.zeros_like(u_on), u_ss.expand_as(u_on)))
        sp = torch.where(active, torch.where(on, s_on, s_off), torch.where(on & switching[None, :], torch.zeros_like(s_on), s_ss.expand_as(s_on)))
        sz = size[b:b + blk, None]
        cu = torch.poisson(0.3 * sz * u.clamp_(min=0.0), generator=gen).clamp_(max=65535)
        cs = torch.poisson(sz * sp.clamp_(min=0.0), generator=gen).clamp_(max=65535)
        sumU[b:b + blk], sumS[b:b + blk] = cu.sum(1).double(), cs.sum(1).double()
        U[b:b + blk, :G] = cu.to(torch.int32).to(torch.int16)
        S[b:b + blk, :G] = cs.to(torch.int32).to(torch.int16)
    fS = sumS.mean() / sumS.clamp(min=1.0)
    fU = sumU.mean() / sumU.clamp(min=1.0)
    cS, cU = ops.CountMatrix(S, G).narrowed(), ops.CountMatrix(U, G).narrowed()
    if cS.t.dtype != cU.t.dtype:
        cS, cU = ops.CountMatrix(S, G), ops.CountMatrix(U, G)
    L = torch.empty((C, G), dtype=torch.float32, device=dev)
    for b in range(0, C, blk):
        L[b:b + blk] = torch.log2(cS.as_int32(b, b + blk).float() * fS[b:b + blk, None].float() + 1.0)
    L -= L.mean(0, keepdim=True)
    torch.manual_seed(SURVEY_SEED0 + cfg)
    Uu, Ss, _ = torch.svd_lowrank(L, q=P, niter=2)
    pcs = (Uu * Ss).double().contiguous()
    del L
    return cS, cU, fS, fU, pcs


def make_dataset(a, dev):
    """The headline dataset of this run (--generator)."""
    if getattr(a, "generator", "survey") == "survey":
        return synth_counts_survey(a.cells, a.genes, a.pca_dims, dev, cfg=3)
    return synth_counts(a.cells, a.genes, a.pca_dims, dev)


def synth(C, G, P, dev, seed=20180811):
    """The same dataset as size-normalised f32 matrices S_sz, U_sz (cells-major) + pcs."""
    cS, cU, fS, fU, pcs = synth_counts(C, G, P, dev, seed)
    S, U = cS.to_float(torch.float32), cU.to_float(torch.float32)
    S.t.mul_(fS[:, None].float())
    U.t.mul_(fU[:, None].float())
    return S, U, pcs


def sample_neighbors_device(embedding, n_neighbors, sampled_fraction, dev, seed=15071990):
    """Embedding kNN (HIP kernel) + weighted subsampling without replacement (analysis.py:1547-1572) on the device.
    The reference draws with numpy's legacy RNG on the host; for the benchmark the draw is an input."""
    from velocyto_amd import ops
    idx, _ = ops.knn_search(embedding, n_neighbors + 1, include_self=False)
    n1 = n_neighbors + 1
    p = torch.linspace(0.5, 0.1, n1, device=dev, dtype=torch.float64)
    p = p / p.sum()
    gen = torch.Generator(device=dev).manual_seed(seed)
    # Efraimidis-Spirakis: the m largest u^(1/p) are a weighted sample without replacement
    keys = torch.log(torch.rand((idx.shape[0], n1), generator=gen, device=dev, dtype=torch.float64)) / p[None, :]
    m = int(sampled_fraction * n1)
    sel = torch.topk(keys, m, dim=1).indices
    # the order of a cell's sampled neighbours carries no meaning: keep each row sorted by neighbour index, so that column
    # tiles of wide lists (nrndm > 256) cover the same index ranges for adjacent cells
    return torch.sort(torch.gather(idx, 1, sel), dim=1).values.contiguous(), idx


class Pipeline:
    """One rank's view of the path.  `dtype`: torch.float32 (production) or torch.float64 (the reference's arithmetic);
    `data`: the synthetic dataset (cS, cU, fS, fU, pcs) shared by the pipelines of one process."""

    def __init__(self, args, dev, rank, world, dtype=torch.float32, data=None, counts="auto"):
        from velocyto_amd import ops, distributed
        self.ops, self.D = ops, distributed
        self.a, self.dev, self.rank, self.world, self.dtype = args, dev, rank, world, dtype
        self.rules = None                        # stage-D branch rule, decided on the first pooled matrix
        C, G = args.cells, args.genes
        # resident inputs: the loom's count layers + per-cell size factors (S_sz = fS * S is never materialised)
        self.cS, self.cU, self.fS, self.fU, self.pcs = data if data is not None else make_dataset(args, dev)
        if counts == "u16":                      # real looms are uint16 on disk (constants.py:11); a few heavy genes exceed 255
            widen = lambda m: m if m.t.dtype == torch.int16 else ops.CountMatrix(m.t.to(torch.int16), m.G)
            self.cS, self.cU = widen(self.cS), widen(self.cU)
        if world > 1:
            dist.broadcast(self.pcs, 0)          # every rank must relabel / shard by the same embedding, bit for bit
        self.collect = world > 1 or distributed.FORCE
        if self.collect:
            # cell-sharded run: relabel the cells along a space-filling curve of the embedding so that a rank's contiguous block
            # of cells is spatially coherent and most sampled neighbours are rank-local (dataset preprocessing, untimed)
            perm = (ops.hilbert_order(self.pcs[:, :2].contiguous()) if args.curve == "hilbert" else ops.morton_order(self.pcs[:, :2].contiguous(), 2)).long()
            self.perm = perm
            self.cS = ops.CountMatrix(self.cS.t.index_select(0, perm).contiguous(), G)
            self.cU = ops.CountMatrix(self.cU.t.index_select(0, perm).contiguous(), G)
            self.fS, self.fU, self.pcs = self.fS[perm].contiguous(), self.fU[perm].contiguous(), self.pcs[perm].contiguous()
        self.space = self.pcs[:, :args.pca_dims].contiguous()
        emb = self.pcs[:, :2].contiguous()
        self.neigh, _ = sample_neighbors_device(emb, args.n_neighbors, args.sampled_fraction, dev)
        self.nrndm = int(self.neigh.shape[1])
        self.c0, self.c1 = distributed.shard_bounds(C, world, rank)
        nloc = self.c1 - self.c0
        self.neigh_loc = self.neigh[self.c0:self.c1].contiguous()
        curve = ops.hilbert_order if args.curve == "hilbert" else (lambda pts: ops.morton_order(pts, 2))
        self.order = curve(emb[self.c0:self.c1]) if args.order == "embedding" else None
        # pooling schedule: a space-filling curve over the leading PCs of the kNN space (locality sort, results unchanged)
        self.pool_order = ((ops.hilbert_order(self.space[self.c0:self.c1]) if args.curve == "hilbert" else ops.morton_order(self.space[self.c0:self.c1], 3))
                           if args.order == "embedding" else None)
        # persistent outputs
        self.Ux_loc = ops.CellMatrix.empty(nloc, G, dtype)
        self.plan = None
        self.sched = None
        self.neigh_k = self.neigh_loc          # neighbour lists in the row numbering of the buffer stage D reads (`e_rows`)
        if self.collect and args.exchange == "halo":
            # SHARDED e: a rank holds its own rows of e = Sx_sz followed by the halo rows its neighbour lists reference
            # (compact buffer of n_loc + n_halo rows; the lists are renumbered once per graph)
            need = torch.zeros(C, dtype=torch.bool, device=dev)
            need[self.neigh_loc.reshape(-1).long()] = True
            need[self.c0:self.c1] = True
            self.plan = distributed.HaloPlan(need, C)
            self.e_rows = ops.CellMatrix(torch.zeros((nloc + self.plan.n_recv, ops.padded_ld(G)), dtype=dtype, device=dev), G)
            self.Sx_loc = self.e_rows.rows(0, nloc)
            self.neigh_k = self.plan.localize(self.neigh_loc)
            if args.overlap:
                # interior cells (all sampled neighbours rank-local) need no remote row: their stage D runs while the halo moves
                base = self.order.long() if self.order is not None else torch.arange(nloc, device=dev)
                inter = ((self.neigh_loc >= self.c0) & (self.neigh_loc < self.c1)).all(1)
                # the first launch takes whole rounds of the device only (distributed.overlap_schedules): two launches, ONE tail
                per_round = torch.cuda.get_device_properties(dev).multi_processor_count * (8 if dtype == torch.float32 else 6)
                self.sched = distributed.overlap_schedules(base, inter, per_round)
                self.n_interior = int(inter.sum())
            self.e_cell0 = 0
        elif self.collect:
            # all-gather exchange: the full-height buffer, own rows written in place by the pooling
            self.e_rows = ops.CellMatrix(torch.zeros((C, ops.padded_ld(G)), dtype=dtype, device=dev), G)
            self.Sx_loc = self.e_rows.rows(self.c0, self.c1)
            self.e_cell0 = self.c0
        else:
            self.Sx_loc = ops.CellMatrix.empty(nloc, G, dtype)
            self.e_rows = self.Sx_loc
            self.e_cell0 = 0
        self.corr_loc = torch.empty((nloc, self.nrndm), dtype=dtype, device=dev)
        self.corr = torch.empty((C, self.nrndm), dtype=dtype, device=dev) if self.collect else self.corr_loc
        self.ev = [torch.cuda.Event(enable_timing=True) for _ in range(12)]
        self.stage_ms = np.zeros(7)
        self.d_ms = []
        # shader clock WHILE stage D runs, measured in this run (ops.ClockProbe: one wave per XCD on a side stream reads the shader-clock
        # counter against the 100 MHz counter every 2 ms; it starts when the main stream reaches stage D and ends before D does).  The
        # probe's eight workgroups take eight CUs away from the 160 KB-LDS workgroups of stage D (236 against 226 ms with it): it runs in
        # EXTRA, untimed steps right after the timed ones (step(probe=True)), never inside the timed region
        self.probe = ops.ClockProbe() if (world == 1 and os.environ.get("VCY_NO_PROBE") != "1") else None
        self.probe_samples = []

    def step(self, timed=False, probe=False):
        ops, a = self.ops, self.a
        C, G, k = a.cells, a.genes, a.k
        c0, c1 = self.c0, self.c1
        nloc = c1 - c0
        ev = self.ev
        ev[0].record()
        # ---- A: kNN graph (analysis.py:1005) -> connectivity weights (:1006-1010) -> pooling (:1012-1013)
        idx, dist_ = ops.knn_search(self.space, k, include_self=False, q0=c0, Q=nloc)
        ev[9].record()
        conn = (dist_ > 0).to(self.dtype)                                      # (knn > 0): zero-distance neighbours drop out
        wrow = torch.cat([torch.ones((nloc, 1), device=self.dev, dtype=self.dtype), conn], 1)     # diag = 1
        wrow = wrow / wrow.sum(1, keepdim=True)
        indices = torch.cat([torch.arange(c0, c1, device=self.dev, dtype=torch.int32)[:, None], idx], 1)
        indptr = torch.arange(0, (nloc + 1) * (k + 1), k + 1, device=self.dev, dtype=torch.int64)
        indices, wrow = ops.canonical_graph_rows(indices, wrow)                 # rows by cell number: scipy's order in the reference
        ev[10].record()
        ops.knn_pool_counts(self.cS, self.cU, self.fS, self.fU, indptr, indices, wrow, dtype=self.dtype, cell0=c0, C_out=nloc,
                            out=self.Sx_loc, out2=self.Ux_loc, validate=False, order=self.pool_order, slab_genes=self.a.slab)
        ev[1].record()
        # ---- B: fit_slope (estimation.py:267-279); sharded: all-reduce of the per-gene moments
        mom = ops.fit_slope_moments(self.Ux_loc, self.Sx_loc)
        self.D.all_reduce_sum(mom)
        gamma = ops.fit_slope_from_moments(mom)
        gamma[~torch.isfinite(gamma)] = 0.0          # fit_gammas' policy for genes without signal (analysis.py:1260)
        ev[2].record()
        # ---- C: predict_U -> velocity -> shift -> signed-sqrt dmat.  Default: folded into stage D's staging of d[c]
        #         (vcy_coldeltacor_partial_fused, bit-identical); --no-fuse materialises dmat with k_velocity_chain.
        dmat = None
        if not a.fuse:
            dmat = ops.velocity_chain(self.Sx_loc, self.Ux_loc, gamma, None, want=("dmat",), transform=ops.SQRT, psc=1e-10)["dmat"]
        ev[3].record()
        if probe and self.probe is not None and self.d_ms:        # (the timed steps have told how long D takes)
            self.probe.stream.wait_event(ev[3])
            self.probe.start(0.9 * self.d_ms[-1])
            self.probe_samples.append(self.probe.samples)
        # ---- D: colDeltaCorSqrtpartial; sharded: every rank needs the rows of e = Sx_sz its neighbour lists reference
        if self.rules is None:                   # decided once, on the first pooled matrix, from whole-matrix reductions all-reduced over the
            #                                      ranks (one host sync; every rank takes the same decision; see ops.partial_rules_for)
            self.rules = ops.RULES_PARTIAL if a.literal_rule else ops.partial_rules_for(
                self.Sx_loc, ops.SQRT, 1e-10, stats=self.D.all_reduce_abs_stats(ops.abs_stats(self.Sx_loc)), cells=C)
        rules = self.rules

        def stage_d(order):
            if a.fuse:
                ops.coldeltacor_partial_fused(self.e_rows, self.Ux_loc, gamma, None, self.neigh_k, ops.SQRT, rules, 1e-10,
                                              cell0=self.e_cell0, u_row0=self.e_cell0, order=order, out=self.corr_loc, validate=False)
            else:
                ops.coldeltacor_partial(self.e_rows, dmat, self.neigh_k, ops.SQRT, rules, 1e-10, cell0=self.e_cell0,
                                        d_row0=self.e_cell0, order=order, out=self.corr_loc, validate=False)
        if self.sched is not None:
            handle = self.plan.begin(self.Sx_loc.t, recv_out=self.e_rows.t[nloc:])    # halo rows packed, all_to_all_single started (async on RCCL)
            ev[4].record()
            stage_d(self.sched[0])                                      # interior cells: overlaps with the transfer
            ev[7].record()
            self.plan.end(handle, self.e_rows.t, row0=nloc)             # stream waits for the transfer, rows land behind the rank's own
            ev[8].record()
            stage_d(self.sched[1])                                      # cells with at least one remote neighbour
        else:
            if self.plan is not None:
                self.plan.end(self.plan.begin(self.Sx_loc.t, recv_out=self.e_rows.t[nloc:]), self.e_rows.t, row0=nloc)      # halo rows only
            elif self.collect:
                self.D.all_gather_rows(self.Sx_loc.t, C, out=self.e_rows.t)
            ev[4].record()
            ev[7].record()
            ev[8].record()
            stage_d(self.order)
        ev[5].record()
        if self.collect:
            self.D.all_gather_rows(self.corr_loc, C, out=self.corr)
        ev[6].record()
        if timed:
            torch.cuda.synchronize()
            t_d = ev[4].elapsed_time(ev[7]) + ev[8].elapsed_time(ev[5])          # both parts of stage D (one part when not overlapped)
            self.stage_ms += np.array([ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), ev[2].elapsed_time(ev[3]),
                                       ev[3].elapsed_time(ev[4]) + ev[7].elapsed_time(ev[8]) + ev[5].elapsed_time(ev[6]), t_d,
                                       ev[0].elapsed_time(ev[9]), ev[10].elapsed_time(ev[1])])
            self.d_ms.append(t_d)
        self.last_gamma = gamma
        return gamma

    def stage_d_clock(self):
        """(mean, min, max) GHz of the shader clock over the probed stage-D launches of this pipeline, or None."""
        if not self.probe_samples:
            return None
        torch.cuda.synchronize()
        f = []
        for smp in self.probe_samples:
            x = smp.cpu().numpy().astype(np.float64)
            dc, dr = np.diff(x[:, :, 0], axis=1), np.diff(x[:, :, 1], axis=1)
            f.append(dc / np.maximum(dr, 1.0) * 0.1)
        per_xcd = np.concatenate(f, axis=1)                     # (8 probe workgroups = 8 XCDs, readings)
        f = per_xcd.ravel()
        return {"mean": float(f.mean()), "min": float(f.min()), "max": float(f.max()), "launches": len(self.probe_samples), "readings": int(f.size),
                "per_xcd_mean": [round(float(v), 4) for v in per_xcd.mean(1)], "per_xcd_min": [round(float(v), 4) for v in per_xcd.min(1)]}

    def probe_clock(self, steps=3):
        """`steps` extra, untimed steps with the clock probe running beside their stage-D launches (after the timed steps: same data, same
        launches, the caches and the power state they left)."""
        if self.probe is None or not self.d_ms:
            return
        for _ in range(steps):
            self.step(probe=True)
        torch.cuda.synchronize()

    def time_dual(self, reps=3):
        """Stage D with the randomised control of estimate_transition_prob (analysis.py:1539-1542): one dual-control launch
        against one single launch, same inputs.  The control matrix here is a stand-in with the right shape and statistics
        (rows of a materialised dmat shuffled over cells with random signs); its values do not change the cost."""
        ops = self.ops
        gamma = self.last_gamma
        dm = ops.velocity_chain(self.Sx_loc, self.Ux_loc, gamma, None, want=("dmat",), transform=ops.SQRT, psc=1e-10)["dmat"]
        gen = torch.Generator(device=self.dev).manual_seed(7)
        prm = torch.randperm(dm.C, generator=gen, device=self.dev)
        sgn = (torch.randint(0, 2, (1, dm.ld), generator=gen, device=self.dev) * 2 - 1).to(self.dtype)
        d_r = ops.CellMatrix((dm.t.index_select(0, prm) * sgn).contiguous(), dm.G)
        del dm
        out_r = torch.empty_like(self.corr_loc)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        def run(dual):
            best = 1e30
            for _ in range(reps):
                e0.record()
                if dual:
                    ops.coldeltacor_partial_fused_dual(self.e_rows, self.Ux_loc, gamma, None, d_r, self.neigh_k, ops.SQRT, self.rules, 1e-10,
                                                       cell0=self.e_cell0, u_row0=self.e_cell0, order=self.order, out=self.corr_loc, out_rndm=out_r, validate=False)
                else:
                    ops.coldeltacor_partial_fused(self.e_rows, self.Ux_loc, gamma, None, self.neigh_k, ops.SQRT, self.rules, 1e-10,
                                                  cell0=self.e_cell0, u_row0=self.e_cell0, order=self.order, out=self.corr_loc, validate=False)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            return best
        single, dual = run(False), run(True)
        out = {"D_single_ms": single, "D_dual_ms": dual, "dual_over_single": dual / single,
               "note": "estimate_transition_prob(calculate_randomized=True): real + randomised-control correlations from one "
                       "vcy_coldeltacor_partial_fused_dual launch; two launches would cost 2.0 x"}
        if self.rules == ops.RULES_PARTIAL_NOPSC:
            # the same launch with the literal branch rule (pseudocount kept): its cost, and how far ALL correlations move
            fast = self.corr_loc.clone()
            e0.record()
            lit = ops.coldeltacor_partial_fused(self.e_rows, self.Ux_loc, gamma, None, self.neigh_k, ops.SQRT, ops.RULES_PARTIAL, 1e-10,
                                                cell0=self.e_cell0, u_row0=self.e_cell0, order=self.order, validate=False)
            e1.record()
            torch.cuda.synchronize()
            ok = torch.isfinite(lit)
            out["literal_rule"] = {"D_single_ms": e0.elapsed_time(e1), "max_abs_dcorr_all_pairs": float((lit[ok] - fast[ok]).abs().max()),
                                   "nan_pattern_equal": bool(torch.equal(torch.isnan(lit), torch.isnan(fast))),
                                   "note": "stage D with VCY_RULES_PARTIAL (sign(t) sqrt(|t| + psc), speedboosted.pyx:372-378) instead of the "
                                           "no-pseudocount form the timed run uses on f32 matrices (ops.partial_rules_for)"}
        return out


def hip_subproblem(pipe, args, Cs, ixs, dtype, rules):
    """The HIP path on the closed sub-problem the CPU baseline runs (first `Cs` cells, all genes): kNN graph among those cells,
    pooling from the count layers, fit_slope, velocity chain folded into the stage-D launch.  Returns (Sx, gamma, corr)."""
    ops = pipe.ops
    dev, G, k = pipe.dev, args.genes, min(args.k, Cs - 1)
    cS = ops.CountMatrix(pipe.cS.t[:Cs].contiguous(), G)
    cU = ops.CountMatrix(pipe.cU.t[:Cs].contiguous(), G)
    fS, fU = pipe.fS[:Cs].contiguous(), pipe.fU[:Cs].contiguous()
    idx, dist_ = ops.knn_search(pipe.space[:Cs].contiguous(), k, include_self=False)
    conn = (dist_ > 0).to(dtype)
    wrow = torch.cat([torch.ones((Cs, 1), device=dev, dtype=dtype), conn], 1)
    wrow = wrow / wrow.sum(1, keepdim=True)
    indices = torch.cat([torch.arange(Cs, device=dev, dtype=torch.int32)[:, None], idx], 1)
    indptr = torch.arange(0, (Cs + 1) * (k + 1), k + 1, device=dev, dtype=torch.int64)
    indices, wrow = ops.canonical_graph_rows(indices, wrow)
    Sx, Ux = ops.knn_pool_counts(cS, cU, fS, fU, indptr, indices, wrow, dtype=dtype, validate=False)
    gamma = ops.fit_slope_from_moments(ops.fit_slope_moments(Ux, Sx))
    gamma[~torch.isfinite(gamma)] = 0.0
    corr = ops.coldeltacor_partial_fused(Sx, Ux, gamma, None, torch.from_numpy(ixs.astype(np.int32)).to(dev), ops.SQRT, rules, 1e-10, validate=False)
    pipe._last_sub_Ux = Ux
    return Sx, gamma, corr


def _host_genes_major(work, name, M, block=1500):
    """The (genes, cells) fp64 form of a cells-major device matrix as <work>/<name>.npy, written gene block by gene block through a pinned
    buffer (no second host copy).  Returns the memory map."""
    C, G = M.C, M.G
    mm = np.lib.format.open_memmap(os.path.join(work, name + ".npy"), mode="w+", dtype=np.float64, shape=(G, C))
    pin = torch.empty((block, C), dtype=torch.float64, pin_memory=True)
    for g0 in range(0, G, block):
        g1 = min(G, g0 + block)
        pin[:g1 - g0].copy_(M.t[:, g0:g1].t().to(torch.float64))
        mm[g0:g1] = pin[:g1 - g0].numpy()
    return mm


def cpu_full_width(pipe, args):
    """Stage D on the host cores AS THE REFERENCE READS IT (speedboosted.pyx:366-378): e and d are the FULL fp64
    (genes, cells) matrices of the run (12 GB each at 50 000 x 30 000, built once from the device matrices into a RAM-backed directory), the
    neighbour lists are the run's own, so every gather e[j * cols + i] has the stride of all 50 000 columns.
      * the reference's OWN kernel (oracle/_ref): it has no cell range, so it is started over all columns and the rows it has finished are
        counted at two instants before it is stopped (oracle.reference_coldeltacor_rate), at the reference's thread rule cpu_count() / 2
        (estimation.py:27-28);
      * the restatement (oracle/velocyto_oracle.c, the same loop order and scratch) on the first cells of the run, at cpu_count() / 2 and at
        all cores."""
    import shutil
    import tempfile
    import oracle
    import psutil
    ops = pipe.ops
    C, G, nr = args.cells, args.genes, pipe.nrndm
    cores = os.cpu_count() or 1
    need = 2 * C * G * 8 + cores * G * nr * 8 * 2.2 + (8 << 30)          # e, d, the kernels' scratch (A and A - mean per thread), slack
    avail = psutil.virtual_memory().available
    if not os.path.isdir("/dev/shm") or avail < need or shutil.disk_usage("/dev/shm").free < 2 * C * G * 8 + (4 << 30):
        return {"skipped": f"host memory: {avail / 2**30:.0f} GiB available, {need / 2**30:.0f} GiB needed for the full-width matrices and the kernels' scratch"}
    t_all = time.perf_counter()
    # the run's own pooled matrices, gammas and correlations (a fresh pipeline on the same resident inputs: the extra lines freed the first one's buffers)
    pipe = Pipeline(args, pipe.dev, 0, 1, dtype=torch.float64, data=(pipe.cS, pipe.cU, pipe.fS, pipe.fU, pipe.pcs), counts=args.counts)
    pipe.step()
    gamma = pipe.last_gamma
    dmat = ops.velocity_chain(pipe.Sx_loc, pipe.Ux_loc, gamma, None, want=("dmat",), transform=ops.SQRT, psc=1e-10)["dmat"]
    work = tempfile.mkdtemp(prefix="vcy_cpu_", dir="/dev/shm")
    try:
        e = _host_genes_major(work, "e", pipe.Sx_loc)
        d = _host_genes_major(work, "d", dmat)
        del dmat
        t_build = time.perf_counter() - t_all
        ixs = pipe.neigh.cpu().numpy().astype(np.intp)
        half = max(1, cores // 2)
        out = {"e_stride_cells": C, "genes": G, "nrndm": nr, "host_matrices_GB": 2 * C * G * 8 / 1e9, "build_s": t_build}
        # ---- the restatement on the first cells, both thread counts (two cells per thread: a second round shows the imbalance of the first)
        rest = {}
        hip = pipe.corr_loc.double().cpu().numpy()
        for th in (half, cores):
            n = min(C, 2 * th)
            t0 = time.perf_counter()
            cc = oracle.coldeltacor_partial_compact(e, d, ixs, "sqrt", 1e-10, threads=th, c0=0, c1=n)
            dt = time.perf_counter() - t0
            ok = np.isfinite(cc[:n])
            rest[f"{th}_threads"] = {"cells": n, "seconds": dt, "cells_per_s": n / dt,
                                     "max_abs_dcorr_hip_vs_restatement": float(np.abs(cc[:n][ok] - hip[:n][ok]).max()),
                                     "nan_pattern_equal": bool(np.array_equal(np.isnan(cc[:n]), np.isnan(hip[:n])))}
        out["restatement"] = rest
        # ---- the reference's own kernel over all columns, stopped after its second reading
        if oracle.reference_module_path() is not None:
            try:
                out["reference_kernel"] = {**oracle.reference_coldeltacor_rate(work, ixs, "sqrt", 1e-10, threads=half, t_first=5.0, t_second=17.0),
                                           "kernel": "velocyto/speedboosted.pyx _colDeltaCorSqrtpartial (oracle/_ref), all columns started, rows finished between two readings",
                                           "thread_rule": "cpu_count() / 2 (estimation.py:27-28)"}
            except Exception as ex:                                          # noqa: BLE001
                out["reference_kernel"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
        else:
            out["reference_kernel"] = {"absent": oracle.reference_module_status()[1]}
        out["wall_s"] = time.perf_counter() - t_all
        return out
    finally:
        shutil.rmtree(work, ignore_errors=True)


def cpu_baseline(pipe, args):
    """The CPU side of the path on the GPU box's host cores.

    (1) Stage D - 98 % of the CPU time - at FULL WIDTH (cpu_full_width): the reference's own kernel and the restatement reading the run's
        50 000-column matrices with the run's neighbour lists.  This is `value`.
    (2) A closed sub-problem of `cpu_cells` cells (all genes, same k / nrndm) for everything else: stages A - C by the oracle restatement
        (their per-cell cost enters `value`), stage D by both kernels once more (the e matrix fits the caches here: the figure the earlier
        rounds quoted, kept for comparison), and the HIP path on the SAME sub-problem in its three arithmetic modes: `parity` reports how far
        each is from the fp64 restatement.
    The oracle is the checker and the baseline here, never the thing measured as the product."""
    import oracle
    ops = pipe.ops
    Cs = min(args.cpu_cells, args.cells)
    G = args.genes
    cores = os.cpu_count() or 1
    half = max(1, cores // 2)
    cnt = lambda cm: cm.as_int32(0, Cs).double()
    S = (cnt(pipe.cS) * pipe.fS[:Cs, None]).cpu().numpy().T.copy()     # S_sz in the reference's (G, Cs) layout
    U = (cnt(pipe.cU) * pipe.fU[:Cs, None]).cpu().numpy().T.copy()
    space = pipe.space[:Cs].cpu().numpy()
    rng = np.random.default_rng(0)
    nr = min(pipe.nrndm, Cs - 1)
    ixs = np.stack([rng.choice(Cs, nr, replace=False) for _ in range(Cs)]).astype(np.intp)
    t0 = time.perf_counter()
    _, _, Sx, Ux = oracle.knn_imputation(S, U, space, k=min(args.k, Cs - 1))
    tA = time.perf_counter() - t0
    t0 = time.perf_counter()
    gam = oracle.fit_slope(Ux, Sx)
    tB = time.perf_counter() - t0
    t0 = time.perf_counter()
    gam0 = np.where(np.isfinite(gam), gam, 0.0)                          # fit_gammas' policy (analysis.py:1260), as in the timed path
    _, _, dS, _ = oracle.velocity_chain(Sx, Ux, gam0, None)
    dmat = oracle.delta_transform(Sx, Sx + dS, "sqrt", 1e-10)
    tC = time.perf_counter() - t0
    t0 = time.perf_counter()
    corr = oracle.coldeltacor_partial_compact(Sx, dmat, ixs, "sqrt", 1e-10, threads=cores)
    tD = time.perf_counter() - t0
    # ---- stage D of the closed sub-problem by the REFERENCE'S OWN kernel where oracle/_ref travelled with the snapshot (speedboosted.pyx
    #      built with the reference's flags by oracle/build_ref.py; run in a subprocess), at the reference's thread rule
    ref = None
    if oracle.reference_module_path() is not None:
        try:
            corr_ref, tD_ref = oracle.reference_coldeltacor(Sx, dmat, ixs, "sqrt", 1e-10, threads=half)
            okr = np.isfinite(corr_ref) & np.isfinite(corr)
            ref = {"D_s": tD_ref, "threads": half, "cells_per_s": Cs / tD_ref,
                   "kernel": "velocyto/speedboosted.pyx _colDeltaCorSqrtpartial, built with the reference's flags (oracle/build_ref.py)",
                   "max_abs_dcorr_restatement_vs_reference": float(np.abs(corr_ref[okr] - corr[okr]).max()),
                   "nan_pattern_equal": bool(np.array_equal(np.isnan(corr_ref), np.isnan(corr)))}
        except Exception as e:                                                  # noqa: BLE001 - the baseline falls back to the restatement
            ref = {"error": f"{type(e).__name__}: {e}"[:300]}
    # ---- parity of the HIP path against these numbers, same inputs, three arithmetic modes
    parity = {"pairs": int(Cs * nr), "genes": G, "against": "the fp64 oracle restatement on the same closed sub-problem (all four stages)"}
    ok = np.isfinite(corr)
    for name, dtype, rules in (("f32_nopsc", torch.float32, None), ("f32_literal", torch.float32, ops.RULES_PARTIAL), ("f64", torch.float64, ops.RULES_PARTIAL)):
        rl = rules
        hSx, hg, hc = hip_subproblem(pipe, args, Cs, ixs, dtype, ops.RULES_PARTIAL)          # first call: Sx for the rule decision
        if rl is None:
            rl = ops.partial_rules_for(hSx, ops.SQRT, 1e-10)
            hSx, hg, hc = hip_subproblem(pipe, args, Cs, ixs, dtype, rl)
        hc = hc.double().cpu().numpy()
        hg = hg.double().cpu().numpy()
        sx = hSx.t[:, :G].double().cpu().numpy().T
        pos = gam0 > 0
        parity[name] = {"rule": ops.RULE_NAMES.get(rl, str(rl)),
                        "max_abs_dcorr": float(np.abs(hc[ok] - corr[ok]).max()),
                        "nan_pattern_equal": bool(np.array_equal(np.isnan(hc), ~ok)),
                        "max_rel_dgamma": float((np.abs(hg[pos] - gam0[pos]) / gam0[pos]).max()),
                        "max_rel_dSx": float((np.abs(sx - Sx) / np.maximum(np.abs(Sx), 1e-30))[Sx != 0].max())}
        if name == "f64":
            # stage D alone on IDENTICAL inputs (the HIP path's own pooled matrices and gammas handed to the oracle).  The figure above runs
            # all four stages on both sides: the two poolings sum in different orders, their results differ in the last bit (max_rel_dSx), and
            # the reference's rule is discontinuous at |t| = 1e-16 (speedboosted.pyx:372: A jumps from 0 to +-sqrt(psc) = 1e-5) - on a small
            # closed sub-problem, where pooled neighbourhoods overlap heavily, many genes of a pair sit exactly there
            ux = pipe._last_sub_Ux.t[:, :G].double().cpu().numpy().T
            _, _, dS_h, _ = oracle.velocity_chain(sx, ux, hg, None)
            d_h = oracle.delta_transform(sx, sx + dS_h, "sqrt", 1e-10)
            c_h = oracle.coldeltacor_partial_compact(sx, d_h, ixs, "sqrt", 1e-10, threads=cores)
            ok_h = np.isfinite(c_h)
            parity[name]["stage_D_on_identical_inputs"] = {"max_abs_dcorr": float(np.abs(hc[ok_h] - c_h[ok_h]).max()),
                                                             "nan_pattern_equal": bool(np.array_equal(np.isnan(hc), ~ok_h))}
            del ux, dS_h, d_h, c_h
        del hSx, hg, hc, sx
    closed = {"cells": Cs, "stage_s": {"A": tA, "B": tB, "C": tC, "D_restatement": tD, **({"D_reference": ref["D_s"]} if ref and "D_s" in ref else {})},
              "restatement_threads": cores, "reference_kernel": ref,
              "note": f"closed sub-problem: {Cs} cells x {G} genes, k={min(args.k, Cs - 1)}, nrndm={nr}; its e matrix is {Cs * G * 8 / 1e6:.0f} MB and the lists index "
                      f"{Cs} columns (row stride {Cs * 8} B): kinder to the CPU than the run's {args.cells} columns"}
    abc_per_cell = (tA + tB + tC) / Cs
    # ---- full width
    try:
        full = cpu_full_width(pipe, args) if (args.cells >= 4 * Cs and pipe.world == 1 and pipe.dtype == torch.float64) else {"skipped": "the run is not larger than the closed sub-problem, or not the f64 single-GPU pass"}
    except Exception as e:                                                      # noqa: BLE001
        torch.cuda.synchronize()
        full = {"error": f"{type(e).__name__}: {e}"[:300]}
    rk = full.get("reference_kernel") or {}
    rs = full.get("restatement") or {}
    if "cells_per_s" in rk:
        dps, kind, th = rk["cells_per_s"], "reference", rk["threads"]
        what = (f"stage D by the reference's own Cython kernel over the run's full {args.cells}-column fp64 matrices (stride {args.cells} cells = {args.cells * 8} B between "
                f"the genes of a cell) with the run's neighbour lists, {th} threads = cpu_count() / 2 (estimation.py:27-28): {rk['rows_second'] - rk['rows_first']} cells finished "
                f"between {rk['t_first']:.1f} s and {rk['t_second']:.1f} s of a run over all columns, then stopped")
    elif rs:
        best = max(rs.values(), key=lambda r: r["cells_per_s"])
        dps, kind, th = best["cells_per_s"], "port", [int(k.split("_")[0]) for k, v in rs.items() if v is best][0]
        what = (f"stage D by the oracle restatement (the reference's loop order and scratch) on the first {best['cells']} cells of the run over the full {args.cells}-column "
                f"fp64 matrices, {th} threads")
    else:
        dps = Cs / (ref["D_s"] if ref and "D_s" in ref else tD)
        kind, th = ("reference", ref["threads"]) if ref and "D_s" in ref else ("port", cores)
        what = f"stage D on the CLOSED sub-problem only (full width not run: {full.get('skipped') or full.get('error')})"
    value = 1.0 / (1.0 / dps + abc_per_cell)
    return {"value": value, "unit": "cells/s", "cores": th, "host_cores": cores, "kind": kind,
            "sample": what + f"; stages A-C by the oracle restatement on the closed sub-problem of {Cs} cells ({abc_per_cell * 1e3:.2f} ms per cell, "
                             f"{100.0 * abc_per_cell * value:.1f} % of the CPU time per cell)",
            "stage_D_cells_per_s": dps, "full_width": full, "closed_subproblem": closed, "parity": parity}


def load_counters(dtype="f32"):
    """Per-launch counters of the dominant kernel in `dtype` from the rocprofv3 --pmc passes committed under profiles/ (file named
    in the bench line).  Returns ({}, None) when nothing is there."""
    for path in (COUNTERS_FILE, COUNTERS_FALLBACK):
        try:
            with open(path) as f:
                rec = json.load(f)
        except Exception:
            continue
        if "f32" in rec or "f64" in rec:              # round-3 layout: one record per arithmetic type
            if rec.get(dtype):
                return rec[dtype], path
        elif dtype == "f32":                          # round-2 layout: the f32 kernel only
            return rec, path
    return {}, None


def dominant_roofline(a, pipe, d_ms, dtype):
    """VALU-issue roofline of the stage-D launch of `pipe` (k_cdc_partial_grouped in f32 or f64) from this run's HIP-event time and the
    committed per-pair-chunk instruction count of the same kernel."""
    C, G, nr = a.cells, a.genes, pipe.nrndm
    nloc = pipe.c1 - pipe.c0
    s = 8 if dtype == "f64" else 4
    pair_genes = float(nloc) * nr * G
    chunk = (6 if s == 4 else 8) * 64 * (16 // s)               # genes per chunk: f32 8 cells x 6 vectors, f64 6 cells x 8 vectors
    pair_chunks = float(nloc) * nr * ((G + chunk - 1) // chunk)
    alg_bytes = nloc * ((nr + 2) * G * s + nr * (4 + s))            # (nrndm+2)*G*s + nrndm*(idx+out) per cell, no reuse credited
    cnt, cnt_path = load_counters(dtype)
    if cnt.get("rules") != pipe.rules:                               # the committed counters are of the other branch rule
        cnt = {}
    instr = cnt.get("valu_insts_per_pair_chunk", None)
    default_wl = (C, G, nr, a.k, pipe.world, a.order, a.fuse, a.curve) == (50000, 30000, 250, 30, 1, "embedding", True, "hilbert")
    traffic = a.traffic_bytes if (a.traffic_bytes is not None and dtype == a.dtype) else (cnt.get("hbm_bytes_per_launch") if default_wl else None)
    rule_name = pipe.ops.RULE_NAMES.get(pipe.rules, str(pipe.rules))
    shape = "8 cells, 6 vectors" if s == 4 else "6 cells, 8 vectors"
    roof = {"bound": "valu", "kernel": f"k_cdc_partial_grouped<{'float' if s == 4 else 'double'}, SQRT, rules {pipe.rules}, {shape}> (velocity chain folded in)",
            "branch_rule": rule_name, "dtype": dtype,
            "unit": "Ginstr/s", "peak": VALU_ISSUE_PEAK / 1e9, "avg_launch_ms": d_ms,
            "peak_is": "VALU issue: 1024 SIMD-32 x 2.4 GHz / 2 clocks per wave64 instruction (MI355X_MICROARCH.md); f32 plain ops measure 2.25 clocks" +
                       (f", f64 add / mul / fma {F64_ISSUE_CLK} (profiles/r04_valu_issue_f64.txt) - see frac_of_f64_issue_peak" if s == 8 else "")}
    # the shader clock of THIS run's stage-D launches (Pipeline.stage_d_clock: vcy_clock_probe on a side stream while D runs); everything
    # that comes from the committed rocprofv3 passes of another run is named profile_*
    clk = pipe.stage_d_clock()
    ghz = clk["mean"] if clk else None
    if clk:
        roof.update({"effective_clock_ghz": ghz, "effective_clock": {**clk, "how": "vcy_clock_probe: one wave per XCD reads the shader-clock counter (s_memtime) against "
                     "the 100 MHz counter (s_memrealtime) every 2 ms on a side stream while the stage-D launches of EXTRA, untimed steps run, right after the "
                     "timed ones in the same process (inside the timed steps the probe's workgroups would take CUs from stage D: 236 against 226 ms)"}})
    if instr is not None:
        achieved = instr * pair_chunks / (d_ms * 1e-3)
        roof.update({"achieved": achieved / 1e9, "frac": achieved / VALU_ISSUE_PEAK,
                     "valu_insts_per_launch": instr * pair_chunks, "profile_valu_insts_per_pair_chunk": instr, "counters_from": os.path.relpath(cnt_path, ROOT),
                     "profile_wave_time": cnt.get("wave_time"), "profile_effective_clock_ghz": cnt.get("effective_clock_ghz"),
                     "profile_launch_ms": cnt.get("profiled_launch_ms")})
        if s == 8:
            roof["frac_of_f64_issue_peak"] = achieved / (VALU_ISSUE_PEAK * 2.0 / F64_ISSUE_CLK)
        if ghz:
            # the chip clocks to its power budget (MI355X_MICROARCH.md, DVFS)
            roof["frac_at_effective_clock"] = achieved / (VALU_ISSUE_PEAK * ghz / 2.4)
            if s == 8:
                roof["frac_of_f64_issue_peak_at_effective_clock"] = achieved / (VALU_ISSUE_PEAK * 2.0 / F64_ISSUE_CLK * ghz / 2.4)
    else:
        roof.update({"achieved": None, "frac": None, "counters_from": None})
    mix = MIX_CLK_PER_ELEMENT.get(pipe.rules) if s == 4 else (MIX_CLK_PER_ELEMENT_F64 if pipe.rules == 1 else None)
    if mix:
        mix_floor_ms = pair_genes * mix / 64.0 / (1024 * 2.4e9) * 1e3
        roof.update({"mix_clk_per_element": mix, "mix_floor_ms": mix_floor_ms, "frac_of_mix_floor": mix_floor_ms / d_ms})
        if ghz:
            roof["frac_of_mix_floor_at_effective_clock"] = mix_floor_ms * 2.4 / ghz / d_ms
    roof.update({"traffic": traffic, "traffic_from": ("--traffic-bytes (a --pmc pass of this command)" if (a.traffic_bytes is not None and dtype == a.dtype) else
                                                      (os.path.relpath(cnt_path, ROOT) + " (the profiled launch of the same kernel and workload)") if traffic else None),
                 "hbm_frac_measured": (traffic / (d_ms * 1e-3) / HBM_PEAK) if traffic else None,
                 "algorithmic_bytes_per_launch": alg_bytes, "vs_noreuse_model": alg_bytes / (d_ms * 1e-3) / HBM_PEAK,
                 "note": "VALU-issue-bound: `achieved` = SQ_INSTS_VALU of this kernel (rocprofv3 pass in counters_from, per pair-chunk, scaled "
                         "by this run's exact pair-chunk count) / HIP-event launch time; `frac` is against the issue peak of one plain "
                         "instruction per 2 clocks per SIMD at 2.4 GHz.  Two things keep a correct kernel away from that peak and are reported beside it: "
                         "(1) the mix the arithmetic needs is slower than plain instructions (v_rsq_f32 8 clocks, f64 add / mul / fma 4, v_rsq_f64 12.5): "
                         "`mix_floor_ms` is the issue time of that mix alone at the measured rates (tools/ubench/valu_issue*.hip), `frac_of_mix_floor` = "
                         "mix_floor_ms / launch time; (2) the chip does not hold 2.4 GHz under this load: `effective_clock_ghz` is measured in THIS run while stage D "
                         "executes (`effective_clock.how`), `*_at_effective_clock` are the same fractions at that clock; the instruction count per pair-chunk, "
                         "`profile_wave_time`, `profile_effective_clock_ghz` (GRBM_GUI_ACTIVE per XCD / duration) and `traffic` are of the PROFILED launch "
                         "named in `counters_from` / `traffic_from`, another run of the same binary.  HBM is not the limit: "
                         "`traffic` (2*FETCH_SIZE + WRITE_SIZE of the PMC passes) is `hbm_frac_measured` of 8 TB/s; `vs_noreuse_model` "
                         "is SURVEY 8(d)'s no-reuse byte model over launch time over 8 TB/s (above 1: neighbour rows are shared by the "
                         "cells of a group out of LDS and by adjacent groups out of L2)."})
    return roof


def run(a, rank, local_rank, world):
    if os.environ.get("VCY_SINGLE_DEVICE", "0") == "1":     # multi-process test on a one-GPU box: every rank on device 0, gloo
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or os.environ.get("VCY_FORCE_COLLECTIVES", "0") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:                      # (world 1 with forced collectives; launchers always set it)
            os.environ["MASTER_PORT"] = str(_free_port())
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        backend = os.environ.get("VCY_DIST_BACKEND", "nccl")   # "nccl" = RCCL over xGMI; "gloo" only for the one-GPU logic test
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        assert dist.get_world_size() == world and dist.get_rank() == rank, "process group does not match the launcher's ranks"
        # every collective shape of the sharded path on tiny tensors first (distributed.self_check): a transport that cannot do
        # the uneven all-to-all of the halo exchange is reported by name and the run falls back to --exchange allgather
        from velocyto_amd import distributed as _D
        chk = _D.self_check(dev)
        bad = [n for n, ok in chk.get("agreed", {}).items() if not ok]
        a.self_check = "skipped" if "skipped" in chk else ("all ok: " + ", ".join(_D.SELF_CHECKS) if not bad else "FAILED: " + "; ".join(f"{n} ({chk.get(n)})" for n in bad))
        if bad:
            print(f"[bench rank {rank}] collective self-check: {a.self_check}", file=sys.stderr, flush=True)
            if any(n in bad for n in ("all_gather_rows_equal", "all_gather_rows_ragged", "all_reduce_min")):
                raise RuntimeError(f"collective self-check: {a.self_check} - the sharded path has no fallback for these")
            if a.exchange == "halo":                 # the halo plan needs the uneven all-to-all and the mask all-gather
                a.exchange = "allgather"
                a.self_check += " -> --exchange allgather"
    import velocyto_amd  # noqa: F401
    from velocyto_amd import _lib
    _lib.lib()   # fail loudly if the HIP library is missing
    if a.workload == "cfg5":
        from velocyto_amd import atlas
        res = atlas.bench_main(a, dev, rank, world)
        finish(res, rank)
        return

    dtype = torch.float64 if a.dtype == "f64" else torch.float32
    pipe = Pipeline(a, dev, rank, world, dtype=dtype, counts=a.counts)
    for _ in range(a.warmup):
        pipe.step()

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()
    smi_before = smi_sample() if rank == 0 else None
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        pipe.step(timed=True)
    barrier()
    dt = time.perf_counter() - t0
    smi_after = smi_sample() if rank == 0 else None
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / a.steps * 1e3
    pipe.probe_clock()                                      # untimed: the shader clock under stage D, for roofline.effective_clock_ghz
    # every rank's own view of the pass (stage times, shard and halo sizes) -> rank 0, for config.parallelism_detail
    per_rank = None
    if dist.is_initialized() and pipe.collect:
        mine = torch.tensor([*(pipe.stage_ms / a.steps), float(pipe.c1 - pipe.c0), float(pipe.plan.n_recv if pipe.plan is not None else 0),
                             float(pipe.plan.n_send if pipe.plan is not None else 0),
                             float(pipe.sched[0].numel() if pipe.sched is not None else 0), float(getattr(pipe, "n_interior", 0))], dtype=torch.float64)
        if dist.get_backend() == "gloo":
            rows = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
            dist.all_gather(rows, mine)
            per_rank = torch.stack(rows).numpy()
        else:
            rows = torch.zeros((dist.get_world_size(), mine.numel()), dtype=torch.float64, device=dev)
            dist.all_gather_into_tensor(rows, mine.to(dev))
            per_rank = rows.cpu().numpy()

    if rank == 0 and a.dump:
        # in a sharded run the cells were relabelled (curve order of the embedding): report in the original labels
        perm = getattr(pipe, "perm", None)
        corr, neigh = pipe.corr, pipe.neigh
        np.savez(a.dump, gamma=pipe.last_gamma.cpu().numpy(), corr=corr.cpu().numpy(), neigh=neigh.cpu().numpy(),
                 perm=(perm.cpu().numpy() if perm is not None else np.arange(a.cells)))
    res = None
    if rank == 0:
        C, G, nr = a.cells, a.genes, pipe.nrndm
        nloc = pipe.c1 - pipe.c0
        s = 8 if a.dtype == "f64" else 4
        d_ms = float(np.mean(pipe.d_ms))
        stage = pipe.stage_ms / a.steps
        roof = dominant_roofline(a, pipe, d_ms, a.dtype)
        rule_name = pipe.ops.RULE_NAMES.get(pipe.rules, str(pipe.rules))
        cbytes = 1 if pipe.cS.t.dtype == torch.uint8 else 2
        pool_bytes = nloc * 2 * (G * cbytes + G * s)
        stages = {
            "A_knn_search": {"ms": stage[5], "bound": "valu", "note": f"exact kNN of {nloc} queries among {C} points in {a.pca_dims} dims; "
                             f"{2.0 * nloc * C * a.pca_dims / (stage[5] * 1e-3) / 1e12:.2f} Tflop/s of distance FMAs (f32 vector peak 157)"},
            "A_pooling": {"ms": stage[6], "bound": "hbm", "algorithmic_bytes": pool_bytes, "achieved_GBs": pool_bytes / (stage[6] * 1e-3) / 1e9,
                          "frac": pool_bytes / (stage[6] * 1e-3) / HBM_PEAK,
                          "note": f"both layers: counts ({cbytes} B) read once + pooled matrix ({s} B) written once per cell and gene"},
            "B_fit_slope": {"ms": stage[1], "bound": "hbm", "algorithmic_bytes": nloc * 2 * G * s, "achieved_GBs": nloc * 2 * G * s / (stage[1] * 1e-3) / 1e9,
                            "frac": nloc * 2 * G * s / (stage[1] * 1e-3) / HBM_PEAK},
        }
        rccl_ranks = dist.get_world_size() if (dist.is_initialized() and dist.get_backend() == "nccl") else 0
        if world > 1 and os.environ.get("VCY_DIST_BACKEND", "nccl") == "nccl":
            assert rccl_ranks == a.gpus, f"--gpus {a.gpus} but RCCL runs {rccl_ranks} ranks"       # never print a multi-GPU line RCCL did not carry
        res = {
            "metric": "cells/sec through knn_imputation->fit_slope->colDeltaCor, 50k cells x 30k genes",
            "value": C / (ms_per_step * 1e-3), "unit": "cells/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "rccl_ranks": rccl_ranks, "collective_self_check": getattr(a, "self_check", "not run (one rank)"),
            "config": {"workload": f"synthetic {C} cells x {G} genes (BASELINE.json configs[2]): knn_imputation(k={a.k}, "
                                   f"{a.pca_dims} PCs) -> fit_slope -> velocity chain -> colDeltaCorSqrtpartial(nrndm={nr}, "
                                   f"n_neighbors={a.n_neighbors}, sampled_fraction={a.sampled_fraction}, psc=1e-10)",
                       "cells": C, "genes": G, "k": a.k, "nrndm": nr,
                       "generator": ("survey: SURVEY.md 8(d) - numpy PCG64(20180808 + 3) parameters, 12 clusters on a branching latent time, closed-form splicing ODE; "
                                     "Poisson draws on the device" if a.generator == "survey" else
                                     "bench: device-side torch generator, seed 20180811 (= 20180808 + cfg 3), alpha ~ LogNormal(0, 1), gamma ~ LogNormal(-0.5, 0.5), 60 % switching "
                                     "genes, one branch point, size ~ LogNormal(0, 0.3), U ~ Poisson(0.3 size u), S ~ Poisson(size s) - the dataset of rounds 1-5; "
                                     "--generator survey is SURVEY.md 8(d)'s 12-cluster tree (DESIGN.md section 4)"),
                       "inputs": f"spliced/unspliced count layers ({'uint8: narrowed, no count of this dataset exceeds 255' if pipe.cS.t.dtype == torch.uint8 else 'uint16, the loom type of constants.py:11'}) "
                                 "+ per-cell size factors (S_sz = factor*counts), pcs, sampled neighbours",
                       "count_layer_dtype": "uint8" if pipe.cS.t.dtype == torch.uint8 else "uint16",
                       "parallelism": "single GPU" if world == 1 else f"cells sharded over {world} GPUs in embedding ({a.curve} curve) order; RCCL "
                                      "all-reduce of fit moments, " + (f"halo exchange of Sx rows into a compact own+halo buffer (all_to_all, {pipe.plan.n_recv} "
                                      f"of {C} rows received by rank 0" + (f"; overlapped with stage D of {int(pipe.sched[0].numel())} of the {pipe.n_interior} interior cells of {nloc}: whole device rounds)"
                                                              if pipe.sched is not None else ")") if pipe.plan is not None else "all-gather of Sx shards") +
                                      ", all-gather of correlation rows",
                       "stage_ms": {"A_knn_imputation": stage[0], "B_fit_slope": stage[1], "C_velocity_chain": stage[2],
                                    "D_exchange": stage[3], "D_coldeltacor": stage[4]},
                       "A_knn_search_ms": stage[5], "A_pooling_ms": stage[6], "B_fit_slope_ms": stage[1], "D_coldeltacor_ms": stage[4],
                       "arithmetic": "f64 storage and moments, literal branch rule: the reference's (speedboosted.pyx:352-443)" if a.dtype == "f64" else
                                     "f32 storage and lane accumulators (production mode; the reference is f64)",
                       **({"parallelism_detail": parallelism_detail(a, pipe, per_rank, s)} if per_rank is not None else {}),
                       "stage_D_rule": rule_name,
                       "velocity_chain": "folded into the staging of d[c] in the stage-D kernel" if a.fuse else "k_velocity_chain (dmat materialised)",
                       "cell_order_D": a.order + (f" ({a.curve} curve)" if a.order == "embedding" else "")},
            "roofline": roof, "stages": stages,
            "telemetry": {"smi_before_timed_steps": smi_before, "smi_after_timed_steps": smi_after,
                          "note": "rocm-smi of GPU 0 (socket power, temperatures, clocks as the SMI reports them) sampled on the host right before and right "
                                  "after the timed steps; the shader clock UNDER stage D is roofline.effective_clock (per XCD: per_xcd_mean / per_xcd_min)"},
        }
        if world == 1 and not a.no_extra:
            res["precision_modes"], res["extra"] = extra_lines(a, dev, pipe, res)
            # the key numbers of the extra lines as scalars of `config` (the driver's record keeps scalars)
            res["config"].update(extra_scalars(res["extra"], res["precision_modes"]))
        if not a.no_cpu_baseline and world == 1:     # reported on rank 0 at N = 1 only
            res["cpu_baseline"] = cpu_baseline(pipe, a)
    finish(res, rank)


def parallelism_detail(a, pipe, per_rank, s):
    """What every rank did and what every collective moved (one pass), from the ranks' own HIP-event times and plan sizes."""
    C, G, nr = a.cells, a.genes, pipe.nrndm
    ld = pipe.ops.padded_ld(G)
    names = ("A_knn_imputation", "B_fit_slope", "C_velocity_chain", "D_exchange", "D_coldeltacor", "A_knn_search", "A_pooling")
    ranks = []
    for r, v in enumerate(per_rank):
        ranks.append({"rank": r, "cells": int(v[7]), "halo_rows_received": int(v[8]), "halo_rows_sent": int(v[9]), "interior_cells": int(v[11]),
                      "cells_run_while_the_halo_moves": int(v[10]),
                      "stage_ms": {n: float(v[i]) for i, n in enumerate(names)}})
    return {"per_rank": ranks,
            "bytes_per_collective": {
                "B_all_reduce_fit_moments": 3 * G * 8,
                "D_halo_all_to_all_received_per_rank": [int(v[8]) * ld * s for v in per_rank] if a.exchange == "halo" else None,
                "D_all_gather_Sx_total": None if a.exchange == "halo" else C * ld * s,
                "D_all_gather_correlation_rows_total": C * nr * s},
            "note": "stage_ms are each rank's own HIP-event times; D_exchange is the time the rank's stream spent waiting for the halo rows "
                    "(0 when the transfer finished under the interior cells' stage D) plus the all-gather of the correlation rows"}


def _free_port():
    """A TCP port nobody listens on right now (self-launched ranks and forced-collective runs without a launcher): successive
    runs on one box never meet a predecessor's socket in TIME_WAIT."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def smi_sample():
    """Socket power, temperatures and clocks of GPU 0 as the SMI reports them (rocm-smi, read-only; an ordinary user may call it).  A
    213-ms-versus-247-ms pair of runs of the same binary at the same shader clock should explain itself: every bench
    line carries one sample taken before the timed steps and one right after them.  Never raises: a box without the tool says so."""
    import shutil
    import subprocess
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    try:
        r = subprocess.run([exe, "-d", "0", "--showpower", "--showtemp", "--showclocks", "--showperflevel", "--json"], capture_output=True, text=True, timeout=20)
        d = json.loads(r.stdout[r.stdout.index("{"):])
        card = d[sorted(d)[0]]
        keep = {}
        for k, v in card.items():
            kl = k.lower()
            if any(w in kl for w in ("power", "temperature", "sclk", "mclk", "fclk", "socclk", "performance level")):
                keep[k] = v
        return keep or {"error": "rocm-smi returned no power / temperature / clock field", "fields": sorted(card)[:12]}
    except Exception as e:                                                      # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"[:200]}


def _short(p, steps):
    """One untimed + `steps` timed steps of a pipeline (wall clock between device syncs, HIP-event stage times inside)."""
    p.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        p.step(timed=True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    p.probe_clock(2)
    return ms


def wide_list_line(a, dev, pipe):
    """Stage D at the reference's DEFAULT list width (analysis.py:1452-1457: n_neighbors = cells / 5, sampled_fraction = 0.3 =>
    nrndm = int(0.3 * (n_neighbors + 1)); 3000 at 50 000 cells), on the headline pipeline's pooled matrices and gammas, velocity
    chain folded in: one launch in column tiles (ops.coldeltacor_partial_fused), timed by HIP events."""
    ops = pipe.ops
    C = a.cells
    nn = C // 5
    emb = pipe.pcs[:, :2].contiguous()
    wide, _ = sample_neighbors_device(emb, nn, 0.3, dev)
    out = torch.empty((C, wide.shape[1]), dtype=pipe.dtype, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.coldeltacor_partial_fused(pipe.e_rows, pipe.Ux_loc, pipe.last_gamma, None, wide, ops.SQRT, pipe.rules, 1e-10, order=pipe.order, out=out, validate=False)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    pairs = float(C) * wide.shape[1]
    fin = torch.isfinite(out)
    # its own roofline: row sharing is very different at this width (adjacent cells share most of their 10 000 nearest), so the
    # instructions per pair-chunk come from a PMC pass of THIS launch (tools/pmc_wide.sh), not from the nrndm = 250 one
    roof = None
    s = 8 if pipe.dtype == torch.float64 else 4
    if s == 8 and pipe.rules == 1:
        try:
            with open(WIDE_COUNTERS_FILE) as f:
                cnt = json.load(f)["f64_wide"]
        except Exception:
            cnt = None
        G = a.genes
        chunk = 8 * 64 * 2
        pair_chunks = pairs * ((G + chunk - 1) // chunk)
        mix_floor_ms = pairs * G * MIX_CLK_PER_ELEMENT_F64 / 64.0 / (1024 * 2.4e9) * 1e3
        roof = {"bound": "valu", "kernel": "k_cdc_partial_grouped<double, SQRT, rules 1, 6 cells, 8 vectors> in column tiles (velocity chain folded in)",
                "unit": "Ginstr/s", "peak": VALU_ISSUE_PEAK / 1e9, "avg_launch_ms": ms, "mix_floor_ms": mix_floor_ms, "frac_of_mix_floor": mix_floor_ms / ms,
                "algorithmic_bytes_per_launch": C * ((wide.shape[1] + 2) * G * s + wide.shape[1] * (4 + s))}
        roof["vs_noreuse_model"] = roof["algorithmic_bytes_per_launch"] / (ms * 1e-3) / HBM_PEAK
        if cnt and cnt.get("valu_insts_per_pair_chunk"):
            achieved = cnt["valu_insts_per_pair_chunk"] * pair_chunks / (ms * 1e-3)
            roof.update({"achieved": achieved / 1e9, "frac": achieved / VALU_ISSUE_PEAK, "frac_of_f64_issue_peak": achieved / (VALU_ISSUE_PEAK * 2.0 / F64_ISSUE_CLK),
                         "profile_valu_insts_per_pair_chunk": cnt["valu_insts_per_pair_chunk"], "profile_launch_ms": cnt.get("profiled_launch_ms"),
                         "profile_wave_time": cnt.get("wave_time"), "profile_effective_clock_ghz": cnt.get("effective_clock_ghz"),
                         "traffic": cnt.get("hbm_bytes_per_launch"),
                         "hbm_frac_measured": (cnt["hbm_bytes_per_launch"] / (ms * 1e-3) / HBM_PEAK) if cnt.get("hbm_bytes_per_launch") else None,
                         "counters_from": os.path.relpath(WIDE_COUNTERS_FILE, ROOT)})
        else:
            roof.update({"achieved": None, "frac": None, "counters_from": None})
    return {"ms": ms, "launches_timed": 1, "nrndm": int(wide.shape[1]), "n_neighbors": nn, "sampled_fraction": 0.3, "dtype": a.dtype,
            "cells_per_s": C / (ms * 1e-3), "roofline": roof,
            "parity": "tests/test_gpu_fullsize.py::test_fullsize_stage_d_reference_default_list_width_against_the_oracle: 128 whole cells x all 3000 columns x "
                      "30 000 genes against the fp64 oracle in every arithmetic mode (f64 1e-9, f32 5e-5), single and dual-control launches",
            "ns_per_pair": ms * 1e6 / pairs, "ns_per_pair_at_nrndm_250": float(np.mean(pipe.d_ms)) * 1e6 / (float(C) * pipe.nrndm),
            "finite_fraction": float(fin.float().mean()), "max_abs_corr": float(out[fin].abs().max()),
            "note": "estimate_transition_prob's defaults (analysis.py:1452-1457); the compact (cells, 3000) output, one launch"}


def facade_lines(a, dev, data, dtype, passes=2, n_markov=2500):
    """The VelocytoLoom facade (velocyto_amd.analysis, the reference's own method names and defaults) on the headline dataset,
    method by method: wall clock between device syncs, last of `passes` passes (the first pays the allocations).  Gives the
    lines SURVEY.md 8(d) lists beside the headline: default fit_gammas (B), calculate_embedding_shift (E), prepare_markov +
    run_markov (F) - and the facade's own cost of A and D (randomised control included) for comparison with the kernel times."""
    import velocyto_amd as vcy
    cS, cU, fS, fU, pcs = data
    vlm = vcy.analysis.VelocytoLoom.from_arrays(cS, cU, dtype=dtype)
    vlm.pcs = pcs.cpu().numpy()
    vlm.ts = vlm.pcs[:, :2].copy()
    out = {}

    def timed(name, fn, *args, **kw):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(*args, **kw)
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) * 1e3
    for _ in range(passes):
        timed("normalize_ms", vlm.normalize, "both", size=True, log=False)
        timed("A_knn_imputation_ms", vlm.knn_imputation, k=a.k, n_pca_dims=a.pca_dims)
        timed("B_fit_gammas_default_ms", vlm.fit_gammas)
        timed("B_fit_gammas_plain_ms", vlm.fit_gammas, fit_offset=False, weighted=False)
        timed("C_predict_U_ms", vlm.predict_U)
        timed("C_calculate_velocity_ms", vlm.calculate_velocity)
        timed("C_calculate_shift_ms", vlm.calculate_shift)
        timed("C_extrapolate_cell_at_t_ms", vlm.extrapolate_cell_at_t)
        timed("D_estimate_transition_prob_ms", vlm.estimate_transition_prob, hidim="Sx_sz", embed="ts", n_neighbors=a.n_neighbors,
              sampled_fraction=a.sampled_fraction)
        timed("E_calculate_embedding_shift_ms", vlm.calculate_embedding_shift)
        timed("F_prepare_markov_ms", vlm.prepare_markov, 2.0, 4.0)
        timed("F_run_markov_ms", vlm.run_markov, n_steps=n_markov)
    out["F_run_markov_steps"] = n_markov
    out["F_run_markov_ms_per_step"] = out["F_run_markov_ms"] / n_markov
    out["dtype"] = "f64" if dtype == torch.float64 else "f32"
    out["torch_peak_GiB"] = torch.cuda.max_memory_allocated() / 2 ** 30
    out["note"] = ("velocyto_amd.analysis.VelocytoLoom methods with the reference's defaults unless a value is named: knn_imputation(k, n_pca_dims), "
                   "fit_gammas() (maxmin_diag weights, offset), estimate_transition_prob(n_neighbors, sampled_fraction; randomised control in the same "
                   "dual launch, numpy's sampling stream replayed on the host), calculate_embedding_shift() (expression scaling, control), "
                   "prepare_markov(2, 4), run_markov(2500); steady-state pass")
    del vlm
    torch.cuda.empty_cache()
    return out


def cfg2_lines(a, dev, dtype):
    """BASELINE.json configs[1]: synthetic 10 000 cells x 20 000 genes, k = 30, 30 PCs, stages A + B through the
    facade (knn_imputation -> fit_gammas(fit_offset=False, weighted=False) = fit_slope), unbalanced and balanced=True, b_sight=240, b_maxl=120."""
    import velocyto_amd as vcy
    C, G = 10000, 20000
    cS, cU, fS, fU, pcs = (synth_counts_survey(C, G, a.pca_dims, dev, cfg=2) if getattr(a, "generator", "survey") == "survey"
                           else synth_counts(C, G, a.pca_dims, dev, seed=20180810))
    vlm = vcy.analysis.VelocytoLoom.from_arrays(cS, cU, dtype=dtype)
    vlm.pcs = pcs.cpu().numpy()
    vlm.normalize("both", size=True, log=False)
    out = {"cells": C, "genes": G, "k": a.k, "dtype": "f64" if dtype == torch.float64 else "f32"}

    def timed(fn, **kw):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(**kw)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3
    for name, kw in (("unbalanced", {}), ("balanced", dict(balanced=True, b_sight=240, b_maxl=120))):
        tA = tB = float("inf")
        for _ in range(3):                          # the first pass pays the allocations; wall clock between syncs also catches the allocator
            tA = min(tA, timed(vlm.knn_imputation, k=a.k, n_pca_dims=a.pca_dims, **kw))       # returning memory (seconds, once): the best of three
            tB = min(tB, timed(vlm.fit_gammas, fit_offset=False, weighted=False))
        out[name] = {"A_knn_imputation_ms": tA, "B_fit_slope_ms": tB, "cells_per_s": C / ((tA + tB) * 1e-3)}
    out["note"] = "facade calls, wall clock between device syncs, best of three passes; balanced = BalancedKNN(sight_k=240, maxl=120) with the greedy balancing on the host (vcy_balance_knn_host32)"
    del vlm
    torch.cuda.empty_cache()
    return out


def extra_lines(a, dev, pipe, res):
    """Everything beside the headline, same workload, one GPU.  Returns (precision_modes, extra).  Every line is guarded: a
    failure or an exhausted time budget is recorded in the line, never raised - the headline is already measured."""
    ops = pipe.ops
    t_start = time.perf_counter()
    C = a.cells
    main_f64 = a.dtype == "f64"
    nS, nU = pipe.cS.narrowed(), pipe.cU.narrowed()      # the layers as ops.CountMatrix keeps them at upload: uint8 where no count exceeds 255
    if nS.t.dtype != nU.t.dtype:
        nS, nU = pipe.cS, pipe.cU
    data = (nS, nU, pipe.fS, pipe.fU, pipe.pcs)
    extra, modes = {}, {}

    def guarded(name, fn, store=extra):
        spent = time.perf_counter() - t_start
        if spent > a.extra_budget_s:
            store[name] = {"skipped": f"time budget of the extra lines ({a.extra_budget_s:.0f} s) spent"}
            return None
        try:
            t0 = time.perf_counter()
            r = fn()
            if isinstance(r, dict):
                r["wall_s"] = time.perf_counter() - t0
            store[name] = r
            return r
        except Exception as e:                                                  # noqa: BLE001
            torch.cuda.synchronize()
            store[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
            return None

    # ---- (1) stage D with the randomised control (dual launch) and at the reference's default list width, on the headline pipeline
    guarded("randomised_control", lambda: pipe.time_dual(reps=2))
    pipe.step()                                       # time_dual left corr_loc with the same values; keep it simple
    torch.cuda.synchronize()
    corr_main, gamma_main = pipe.corr_loc.clone(), pipe.last_gamma.clone()
    guarded("D_reference_defaults_nrndm3000", lambda: wide_list_line(a, dev, pipe))
    d_main = float(np.mean(pipe.d_ms))
    pipe.Sx_loc = pipe.e_rows = pipe.Ux_loc = None     # make room for the other pipelines and the facade
    torch.cuda.empty_cache()

    # ---- (2) the same pass in the build's other arithmetic modes
    def other_modes():
        other = torch.float32 if main_f64 else torch.float64
        oname = "f32" if main_f64 else "f64"
        po = Pipeline(a, dev, 0, 1, dtype=other, data=data, counts=a.counts)
        n = 3 if main_f64 else 2
        ms = _short(po, n)
        st = po.stage_ms / n
        d_o = float(np.mean(po.d_ms))
        ok = torch.isfinite(po.corr_loc) & torch.isfinite(corr_main)
        dcorr = float((po.corr_loc[ok].double() - corr_main[ok].double()).abs().max())
        g_a, g_b = po.last_gamma.double(), gamma_main.double()
        g64 = g_b if main_f64 else g_a
        dgam = float(((g_a - g_b).abs() / g64.abs().clamp(min=1e-30))[g64 > 0].max())
        rule_o = ops.RULE_NAMES.get(po.rules, str(po.rules))
        cmpd = {"max_abs_dcorr_all_pairs": dcorr, "pairs_compared": int(ok.sum()),
                "nan_pattern_equal": bool(torch.equal(torch.isnan(po.corr_loc), torch.isnan(corr_main))), "max_rel_dgamma": dgam}
        key = "f32_production" if main_f64 else "f64_reference_arithmetic"
        modes[key] = {"dtype": oname, "cells_per_s": C / (ms * 1e-3), "ms_per_step": ms, "steps": n, "warmup": 1, "D_ms": d_o, "stage_D_rule": rule_o,
                      "stage_ms": {"A_knn_imputation": st[0], "A_knn_search": st[5], "A_pooling": st[6], "B_fit_slope": st[1], "C_velocity_chain": st[2], "D_coldeltacor": st[4]},
                      "roofline": dominant_roofline(a, po, d_o, oname), "vs_headline": cmpd,
                      "is": ("f32 storage and lane accumulators, the no-pseudocount form of the rule where ops.partial_rules_for admits it: the build's production "
                             "mode, narrower than the reference's arithmetic - reported, not the headline") if main_f64 else
                            "f64 storage and moments, literal rule: the reference's arithmetic"}
        if other == torch.float32:
            dual = po.time_dual(reps=2)               # f32: dual control, and the literal rule in the same launch shape
            modes[key]["randomised_control"] = {k: dual[k] for k in ("D_single_ms", "D_dual_ms", "dual_over_single")}
            lit = dual.get("literal_rule")
            if lit:
                ms_lit = ms - d_o + lit["D_single_ms"]
                modes["f32_literal_rule"] = {"dtype": "f32", "cells_per_s": C / (ms_lit * 1e-3), "ms_per_step": ms_lit, "D_ms": lit["D_single_ms"],
                                             "stage_D_rule": ops.RULE_NAMES[ops.RULES_PARTIAL],
                                             "max_abs_dcorr_vs_f32_production_all_pairs": lit["max_abs_dcorr_all_pairs"], "nan_pattern_equal": lit["nan_pattern_equal"],
                                             "is": "the f32 run's stages A-C + one stage-D launch with VCY_RULES_PARTIAL (pseudocount kept)"}
            po.step()
            torch.cuda.synchronize()
            if a.counts == "u16" and data[0].t.dtype == torch.uint8:
                c32 = po.corr_loc.clone()
                p8 = Pipeline(a, dev, 0, 1, dtype=torch.float32, data=data, counts="auto")
                ms8 = _short(p8, 2)
                modes["f32_uint8_layers"] = {"dtype": "f32", "ms_per_step": ms8, "cells_per_s": C / (ms8 * 1e-3), "A_pooling_ms": (p8.stage_ms / 2)[6],
                                             "same_results_as_uint16": bool(torch.equal(p8.corr_loc, c32)),
                                             "is": "f32_production with the count layers narrowed to uint8 (ops.CountMatrix.narrowed: lossless, no count of this "
                                                   "dataset exceeds 255)"}
                del p8, c32
        del po
        torch.cuda.empty_cache()
        return None
    guarded("_modes", other_modes, store=modes)
    failed = modes.pop("_modes", None)
    if failed:
        modes["not_measured"] = failed

    def narrowed_layers():
        """The headline pass with the count layers narrowed to uint8 where that is lossless (what the product's upload keeps for this dataset,
        ops.CountMatrix.narrowed), timed with the headline's --steps: `value` is on uint16 layers, the loom's own type (constants.py:11)."""
        if not (a.counts == "u16" and data[0].t.dtype == torch.uint8):
            return {"skipped": "the dataset's layers do not narrow (a count above 255) or the headline already runs on the narrowed layers"}
        p8 = Pipeline(a, dev, 0, 1, dtype=torch.float64 if main_f64 else torch.float32, data=data, counts="auto")
        ms8 = _short(p8, a.steps)
        st = p8.stage_ms / a.steps
        ok = torch.isfinite(p8.corr_loc)
        r = {"dtype": a.dtype, "steps": a.steps, "warmup": 1, "ms_per_step": ms8, "cells_per_s": C / (ms8 * 1e-3), "A_pooling_ms": st[6], "D_ms": float(np.mean(p8.d_ms)),
             "same_results_as_uint16": bool(torch.equal(p8.corr_loc[ok], corr_main[ok]) and torch.equal(torch.isnan(p8.corr_loc), torch.isnan(corr_main))),
             "vs_headline": C / (ms8 * 1e-3) / res["value"],
             "is": "the headline pass with uint8 count layers (1 byte per count gathered by the pooling instead of 2): lossless for this dataset, "
                   "not what a loom delivers"}
        del p8
        torch.cuda.empty_cache()
        return r
    guarded(f"{a.dtype}_uint8_layers", narrowed_layers, store=modes)
    modes["headline"] = {"dtype": a.dtype, "cells_per_s": res["value"], "ms_per_step": res["ms_per_step"], "D_ms": d_main,
                         "stage_D_rule": ops.RULE_NAMES.get(pipe.rules, str(pipe.rules)), "is": "`value`: the K timed steps of this run"}
    del corr_main

    # ---- (3) the facade lines (E, F, default fit_gammas) and cfg2, in the headline arithmetic
    dtype = torch.float64 if main_f64 else torch.float32
    guarded("facade", lambda: facade_lines(a, dev, data, dtype))
    guarded("cfg2", lambda: cfg2_lines(a, dev, dtype))
    extra["wall_s_total"] = time.perf_counter() - t_start
    return modes, extra


def extra_scalars(extra, modes):
    """The extra lines' key numbers, flat (E, F, default fit_gammas, the nrndm = 3000 secondary, cfg2)."""
    out = {}
    fac = extra.get("facade") or {}
    for k in ("E_calculate_embedding_shift_ms", "F_prepare_markov_ms", "F_run_markov_ms_per_step", "F_run_markov_steps", "B_fit_gammas_default_ms",
              "A_knn_imputation_ms", "D_estimate_transition_prob_ms"):
        if k in fac:
            out[("facade_" + k) if k[0] in "AD" else k] = fac[k]
    w = extra.get("D_reference_defaults_nrndm3000") or {}
    if "ms" in w:
        out["D_reference_defaults_nrndm3000_ms"] = w["ms"]
    rc = extra.get("randomised_control") or {}
    if "dual_over_single" in rc:
        out["D_dual_control_over_single"] = rc["dual_over_single"]
    c2 = extra.get("cfg2") or {}
    for name in ("unbalanced", "balanced"):
        if name in c2:
            out[f"cfg2_{name}_A_ms"] = c2[name]["A_knn_imputation_ms"]
            out[f"cfg2_{name}_B_ms"] = c2[name]["B_fit_slope_ms"]
    for key, short in (("f32_production", "f32_production"), ("f32_literal_rule", "f32_literal"), ("f64_reference_arithmetic", "f64"),
                       ("f64_uint8_layers", "f64_uint8_layers"), ("f32_uint8_layers", "f32_uint8_layers")):
        if key in modes and "cells_per_s" in modes[key]:
            out[f"{short}_cells_per_s"] = modes[key]["cells_per_s"]
    return out


def finish(res, rank):
    def flush_c_stdio():                             # RCCL writes its banner through C stdio, which is block-buffered on a pipe:
        sys.stdout.flush()                           # push it out now, so that nothing can land after the JSON line at exit
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
    if dist.is_initialized():
        flush_c_stdio()                              # every rank, before the last barrier
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:                                    # the ONE JSON line, last thing on stdout
        flush_c_stdio()
        print(json.dumps(res), flush=True)


def _spawned(local_rank, a):
    os.environ.update({"RANK": str(local_rank), "LOCAL_RANK": str(local_rank), "WORLD_SIZE": str(a.gpus)})
    run(a, local_rank, local_rank, a.gpus)


def main():
    a = parse()
    if "WORLD_SIZE" in os.environ:                   # launched by torch.distributed.run (the driver's way)
        world, rank, local_rank = int(os.environ["WORLD_SIZE"]), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
        assert world == a.gpus or world == 1, f"--gpus {a.gpus} but WORLD_SIZE={world}"
        run(a, rank, local_rank, world)
    elif a.gpus > 1:                                 # self-launch: one process per GPU
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            os.environ["MASTER_PORT"] = str(_free_port())
        import torch.multiprocessing as mp
        mp.spawn(_spawned, args=(a,), nprocs=a.gpus, join=True)
    else:
        run(a, 0, 0, 1)


if __name__ == "__main__":
    main()
