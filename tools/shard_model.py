#!/usr/bin/env python3
"""Predicted 1 / 2 / 4 / 8-GPU curve of the sharded headline run (BASELINE.json configs[3]) from ONE GPU: what a future
SCALE_r*.json can be judged against.

No multi-GPU node is available to the build.  Everything a rank does between two collectives is ordinary single-GPU work on a
shard, so it is MEASURED here - for every rank of every world size, one after the other on the same device, with the same
kernels, schedules and halo plans the sharded bench.py builds (the relabelled problem, shard bounds of distributed.shard_bounds,
the interior / remote split of stage D) - and only the transfers are MODELLED, from the exact byte counts of the plans and the
xGMI figures of MI355X_MICROARCH.md (7 links x 153 GB/s per GPU, point-to-point: a transfer between two GPUs is bound by their
one link; EFF is the fraction of the link rate a large RCCL message reaches):

    t(N) = max over ranks [ A(kNN of the shard's queries + pooling of the shard) + B(moments of the shard) + t_allreduce
                            + max(D_interior, t_halo) + D_remote + t_allgather_corr ]

usage (GPU box):  python tools/shard_model.py [--cells 50000 --genes 30000]  ->  one JSON document on stdout
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from velocyto_amd import ops, distributed

LINK = 153e9          # B/s per xGMI link and direction (MI355X_MICROARCH.md)
EFF = 0.75            # fraction of the link rate assumed for a large point-to-point message
LAT = 25e-6           # s per collective launch + first byte (small-message all-reduce / all-gather: a few such latencies)

ap = argparse.ArgumentParser()
ap.add_argument("--cells", type=int, default=50000)
ap.add_argument("--genes", type=int, default=30000)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--dtype", choices=["f32", "f64"], default="f64", help="arithmetic of the modelled run (f64 = bench.py's headline)")
a0 = ap.parse_args()
DT = torch.float64 if a0.dtype == "f64" else torch.float32
ES = 8 if a0.dtype == "f64" else 4
sys.argv = ["bench.py", "--cells", str(a0.cells), "--genes", str(a0.genes), "--dtype", a0.dtype]
a = bench.parse()
dev = ops.require_gpu()
C, G, k = a.cells, a.genes, a.k
cS, cU, fS, fU, pcs = bench.synth_counts(C, G, a.pca_dims, dev)
perm = ops.hilbert_order(pcs[:, :2].contiguous()).long()            # the relabelling of the sharded run
cS = ops.CountMatrix(cS.t.index_select(0, perm).contiguous(), G)
cU = ops.CountMatrix(cU.t.index_select(0, perm).contiguous(), G)
fS, fU, pcs = fS[perm].contiguous(), fU[perm].contiguous(), pcs[perm].contiguous()
space, emb = pcs[:, :a.pca_dims].contiguous(), pcs[:, :2].contiguous()
neigh, _ = bench.sample_neighbors_device(emb, a.n_neighbors, a.sampled_fraction, dev)
nr = int(neigh.shape[1])
ld = ops.padded_ld(G)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn):
    best = 1e30
    for _ in range(a0.reps):
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


# the whole pooled matrices once (every rank's e rows and halo rows are rows of these), gamma, the branch rule
idx, dist_ = ops.knn_search(space, k, include_self=False)
conn = (dist_ > 0).to(DT)
wrow = torch.cat([torch.ones((C, 1), device=dev, dtype=DT), conn], 1)
wrow = wrow / wrow.sum(1, keepdim=True)
rows_g = torch.cat([torch.arange(C, device=dev, dtype=torch.int32)[:, None], idx], 1)
rows_g, wrow = ops.canonical_graph_rows(rows_g, wrow)
indptr_all = torch.arange(0, (C + 1) * (k + 1), k + 1, device=dev, dtype=torch.int64)
Sx, Ux = ops.knn_pool_counts(cS, cU, fS, fU, indptr_all, rows_g, wrow, dtype=DT, validate=False)
gamma = ops.fit_slope_from_moments(ops.fit_slope_moments(Ux, Sx))
gamma[~torch.isfinite(gamma)] = 0.0
rules = ops.partial_rules_for(Sx, ops.SQRT, 1e-10)

out = {"workload": {"cells": C, "genes": G, "k": k, "nrndm": nr, "dtype": a0.dtype}, "assumptions": {"xgmi_link_Bps": LINK, "link_efficiency": EFF, "collective_latency_s": LAT},
       "worlds": {}}
for N in (1, 2, 4, 8):
    ranks = []
    for r in range(N):
        c0, c1 = distributed.shard_bounds(C, N, r)
        nloc = c1 - c0
        nl = neigh[c0:c1].contiguous()
        # ---- A: the shard's kNN queries against all cells + pooling of the shard's rows
        t_knn = timed(lambda: ops.knn_search(space, k, include_self=False, q0=c0, Q=nloc))
        ip = indptr_all[: nloc + 1]
        rg, ww = rows_g[c0:c1].contiguous().reshape(-1), wrow[c0:c1].contiguous().reshape(-1)
        Sx_l, Ux_l = ops.CellMatrix.empty(nloc, G, DT), ops.CellMatrix.empty(nloc, G, DT)
        order_p = ops.hilbert_order(space[c0:c1])
        t_pool = timed(lambda: ops.knn_pool_counts(cS, cU, fS, fU, ip, rg, ww, dtype=DT, cell0=c0, C_out=nloc, out=Sx_l, out2=Ux_l,
                                                   validate=False, order=order_p))
        # ---- B
        t_fit = timed(lambda: ops.fit_slope_moments(Ux_l, Sx_l))
        # ---- D: compact buffer [own rows | halo rows ascending], renumbered lists, interior / remote schedules (bench.Pipeline)
        need = torch.zeros(C, dtype=torch.bool, device=dev)
        need[nl.reshape(-1).long()] = True
        need[c0:c1] = False
        halo = torch.nonzero(need).ravel()
        n_halo = int(halo.numel())
        e_rows = ops.CellMatrix(torch.cat([Sx.t[c0:c1], Sx.t.index_select(0, halo)], 0).contiguous(), G)
        ixs = ops.localize_rows(nl, c0, c1, halo)
        base = ops.hilbert_order(emb[c0:c1]).long()
        inter = ((nl >= c0) & (nl < c1)).all(1)
        s_in, s_out = distributed.overlap_schedules(base, inter, torch.cuda.get_device_properties(dev).multi_processor_count * (6 if ES == 8 else 8))
        corr = torch.empty((nloc, nr), dtype=DT, device=dev)
        Ux_r = Ux.rows(c0, c1)
        Ux_r = ops.CellMatrix(Ux_r.t.contiguous(), G)

        def d(order):
            if order is not None and order.numel() == 0:
                return
            ops.coldeltacor_partial_fused(e_rows, Ux_r, gamma, None, ixs, ops.SQRT, rules, 1e-10, cell0=0, u_row0=0, order=order, out=corr, validate=False)
        if N == 1:
            t_in, t_out = timed(lambda: d(base.to(torch.int32))), 0.0
        else:
            t_in, t_out = timed(lambda: d(s_in)), timed(lambda: d(s_out))
        # ---- transfers: bytes per peer link of the halo all-to-all (rows of this rank's need mask owned by each peer)
        per_peer = [int(((halo >= b0) & (halo < b1)).sum()) * ld * ES for (b0, b1) in distributed.all_shard_bounds(C, N)]
        t_halo = (max(per_peer) / (LINK * EFF) + LAT) if N > 1 else 0.0
        t_ar = (2 * LAT + 3 * G * 8 / (LINK * EFF)) if N > 1 else 0.0                     # 720 KB: latency-bound
        t_ag = (LAT + (C - nloc) * nr * ES / (7 * LINK * EFF) * (7.0 / max(1, N - 1))) if N > 1 else 0.0     # (N - 1) peers over their own links
        total = t_knn + t_pool + t_fit + t_ar * 1e3 + max(t_in, t_halo * 1e3) + t_out + t_ag * 1e3
        ranks.append({"rank": r, "cells": nloc, "halo_rows": n_halo, "interior_cells": int(inter.sum()), "cells_in_first_launch": int(s_in.numel()), "halo_bytes_per_peer": per_peer,
                      "ms": {"A_knn_search": t_knn, "A_pooling": t_pool, "B_fit_moments": t_fit, "B_all_reduce_model": t_ar * 1e3,
                             "D_interior": t_in, "D_halo_transfer_model": t_halo * 1e3, "D_remote": t_out, "all_gather_corr_model": t_ag * 1e3, "total": total}})
        del e_rows, Sx_l, Ux_l, corr, Ux_r
    worst = max(ranks, key=lambda x: x["ms"]["total"])
    out["worlds"][str(N)] = {"predicted_ms_per_step": worst["ms"]["total"], "predicted_cells_per_s": C / (worst["ms"]["total"] * 1e-3),
                             "slowest_rank": worst["rank"], "ranks": ranks}
t1 = out["worlds"]["1"]["predicted_ms_per_step"]
for N in ("1", "2", "4", "8"):
    w = out["worlds"][N]
    w["speedup_vs_1"] = t1 / w["predicted_ms_per_step"]
    w["efficiency"] = w["speedup_vs_1"] / int(N)
print(json.dumps(out))
