"""The hot path at atlas scale: knn_imputation -> fit_slope -> velocity chain -> colDeltaCor on SPARSE count layers,
streamed over cell blocks, cells sharded over ranks with NOTHING of cell-count size replicated in dense form.

BASELINE.json configs[4] / SURVEY.md section 8(e), last row: 1M cells x 30k genes, CSR input at ~8 % density.  The reference
cannot run this size: it loads every layer as a dense float64 (G, C) array (analysis.py:59-61; 240 GB per layer) and pools
with a dense x sparse product (neighbors.py:416-423).  The arithmetic below is the reference's, stage by stage, with the
same kernels as the resident path (bench.py `Pipeline`); what changes is where the rows live:

  counts   CSR per rank for its own cells (ops.CsrCounts) plus the COUNT-ROW HALO: the rows of every cell whose pooled
           vector the rank will need (its cells' sampled embedding neighbours) and of those cells' kNN neighbours.  A count
           row is ~12 KB (2400 non-zeros x 5 B) where a pooled row is 120 KB, so ranks exchange count rows ONCE per dataset
           (HaloPlan.fetch_ragged) and afterwards pool every row of e = Sx_sz they read themselves: no dense row ever
           crosses a link, e is sharded by construction, pcs / size factors travel as small vectors.
  A        vcy_knn_pool_csr merges the k + 1 sparse rows of a cell into a dense f32 row (LDS slab per wave), block by block.
  B        fit_slope moments of the rank's own cells, summed over blocks, all-reduced (3 G doubles).
  C + D    per block: e rows = [block | sampled neighbours outside the block], pooled on the spot when the matrices are not
           resident; vcy_coldeltacor_partial_fused on the block with renumbered neighbour lists.

One block that holds all of a rank's cells = the resident mode (Sx, Ux kept between the passes; the layout 8 x 288 GB
affords at 1M cells); smaller blocks trade a second pooling pass for O(block) memory, which lets ONE GPU walk a dataset
whose dense matrices exceed its HBM.  Results do not depend on the block size or on the number of ranks (tests).
"""
from __future__ import annotations

import math
import time
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import distributed as D
from . import ops

KEY_CHUNK = 4096          # cells per deterministic block of the synthetic generator and of the sampling keys
PRUNE_FROM = 100_000      # cells from which the kNN search in the PCA space is projection-pruned (AtlasPath(knn="auto"))


def _keyed_uniform(rows0: int, rows1: int, ncol: int, seed: int, dev) -> torch.Tensor:
    """float64 uniforms for global rows rows0..rows1-1, identical on every rank and for every sharding: row r comes from the
    generator of its KEY_CHUNK-row chunk."""
    out = torch.empty((rows1 - rows0, ncol), dtype=torch.float64, device=dev)
    for ch in range(rows0 // KEY_CHUNK, (rows1 - 1) // KEY_CHUNK + 1):
        gen = torch.Generator(device=dev).manual_seed(int(seed) * 1000003 + ch)
        blk = torch.rand((KEY_CHUNK, ncol), generator=gen, device=dev, dtype=torch.float64)
        a, b = max(rows0, ch * KEY_CHUNK), min(rows1, (ch + 1) * KEY_CHUNK)
        out[a - rows0:b - rows0] = blk[a - ch * KEY_CHUNK:b - ch * KEY_CHUNK]
    return out


def sample_neighbors(emb_full: torch.Tensor, c0: int, c1: int, n_neighbors: int, sampled_fraction: float,
                     sampling_probs=(0.5, 0.1), seed: int = 15071990) -> torch.Tensor:
    """estimate_transition_prob's neighbour subsample (analysis.py:1547-1572) for cells c0..c1-1: the n_neighbors + 1
    nearest cells in the embedding (query excluded), of which int(sampled_fraction * (n_neighbors + 1)) are drawn without
    replacement with probabilities falling linearly from sampling_probs[0] to [1] with the rank (Efraimidis-Spirakis keys;
    the reference draws with numpy's legacy stream on the host - replayed exactly by ops.choice_stream_host in the facade;
    at atlas scale the draw is keyed by the global cell number so that it does not depend on the sharding).  Rows are
    returned sorted by neighbour number (a pair's value does not depend on its column)."""
    dev = emb_full.device
    n1 = n_neighbors + 1
    idx, _ = ops.knn_search(emb_full, n1, include_self=False, q0=c0, Q=c1 - c0)
    p = torch.linspace(sampling_probs[0], sampling_probs[1], n1, device=dev, dtype=torch.float64)
    p = p / p.sum()
    m = int(sampled_fraction * n1)
    out = torch.empty((c1 - c0, m), dtype=torch.int32, device=dev)
    step = 16384
    for s in range(c0, c1, step):
        e = min(c1, s + step)
        keys = torch.log(_keyed_uniform(s, e, n1, seed, dev)) / p[None, :]
        sel = torch.topk(keys, m, dim=1).indices
        out[s - c0:e - c0] = torch.sort(torch.gather(idx[s - c0:e - c0], 1, sel), dim=1).values
    return out.contiguous()


class AtlasPath:
    """One rank's share of the path.  The rank owns the cells c0 .. c0 + cS.C - 1 of a dataset of C_total cells whose labels
    follow a space-filling curve of the embedding, e.g. ops.hilbert_order (so that blocks and shards are spatially coherent:
    a block's sampled neighbours mostly lie inside it and the halos stay small; any labelling gives the same numbers)."""

    def __init__(self, cS: ops.CsrCounts, cU: ops.CsrCounts, fS: torch.Tensor, fU: torch.Tensor, pcs: torch.Tensor, embedding: torch.Tensor, *,
                 c0: int = 0, C_total: Optional[int] = None, k: int = 30, n_neighbors: int = 500, sampled_fraction: float = 0.5,
                 sampling_probs=(0.5, 0.1), block_cells: int = 0, dtype=torch.float32, psc: float = 1e-10, seed: int = 15071990,
                 knn: str = "auto"):
        self.dev = dev = cS.indptr.device
        self.knn_mode = knn
        self.dtype = ops.resolve_dtype(dtype)
        self.G, self.k, self.psc = cS.G, int(k), float(psc)
        self.rules = None                  # ops.partial_rules_for(...) from the all-reduced whole-matrix facts of the first pass
        self.rank, self.world = D.world()
        self.c0, self.nloc = int(c0), cS.C
        self.c1 = self.c0 + self.nloc
        self.C = int(C_total) if C_total is not None else self.nloc
        assert cU.C == self.nloc and cU.G == self.G and fS.numel() == self.nloc and pcs.shape[0] == self.nloc and embedding.shape[0] == self.nloc
        if self.world > 1:
            assert (self.c0, self.c1) == D.shard_bounds(self.C, self.world, self.rank), "cells must be sharded by distributed.shard_bounds"
        self.block_cells = int(block_cells) if block_cells and block_cells > 0 else self.nloc
        # ---- small replicated vectors: the kNN space and the embedding of ALL cells (C x (P + 2) doubles: 256 MB at 1M cells)
        pcs_full = D.all_gather_rows(pcs.double().contiguous(), self.C)
        emb_full = D.all_gather_rows(embedding.double().contiguous(), self.C)
        # ---- kNN graph of the own cells (analysis.py:1005-1010): nearest-first, self excluded; weights (knn > 0) with diag = 1
        self.knn_stats: Dict[str, float] = {}
        idx, dist_ = self._knn(pcs_full)
        conn = (dist_ > 0).to(self.dtype)
        wrow = torch.cat([torch.ones((self.nloc, 1), device=dev, dtype=self.dtype), conn], 1)
        wrow = (wrow / wrow.sum(1, keepdim=True)).contiguous()
        grow = torch.cat([torch.arange(self.c0, self.c1, device=dev, dtype=torch.int32)[:, None], idx], 1)               # global cell numbers
        grow, wrow = ops.canonical_graph_rows(grow, wrow)       # pooled in ascending GLOBAL cell number: scipy's order, and independent of the sharding
        # ---- sampled embedding neighbours of the own cells (global numbers)
        self.neigh = sample_neighbors(emb_full, self.c0, self.c1, n_neighbors, sampled_fraction, sampling_probs, seed)
        self.nrndm = int(self.neigh.shape[1])
        self._pcs_full = pcs_full
        del emb_full
        # ---- E rows: the rows of e = Sx_sz this rank reads = own cells + sampled neighbours owned elsewhere (e_out, ascending)
        need_e = torch.zeros(self.C, dtype=torch.bool, device=dev)
        need_e[self.neigh.reshape(-1).long()] = True
        need_e[self.c0:self.c1] = True
        plan_e = D.HaloPlan(need_e, self.C)
        self.e_out = plan_e.recv_idx.clone()                                          # global numbers, ascending
        grow_e = torch.cat([grow, plan_e.fetch(grow)], 0)                             # graph rows of the E rows, own first
        wrow_e = torch.cat([wrow, plan_e.fetch(wrow)], 0)
        # ---- K rows: the COUNT rows needed to pool every E row = own + everything the E rows' graph rows name
        need_k = torch.zeros(self.C, dtype=torch.bool, device=dev)
        need_k[grow_e.reshape(-1).long()] = True
        need_k[self.c0:self.c1] = True
        plan_k = D.HaloPlan(need_k, self.C)
        self.k_out = plan_k.recv_idx.clone()
        self.n_count_halo, self.n_e_halo = int(self.k_out.numel()), int(self.e_out.numel())

        def with_halo(c: ops.CsrCounts, f: torch.Tensor):
            lens = c.indptr[1:] - c.indptr[:-1]
            lens_h, idx_h, dat_h = plan_k.fetch_ragged(lens, c.indices, c.data)
            ptr = torch.zeros(self.nloc + lens_h.numel() + 1, dtype=torch.int64, device=dev)
            torch.cumsum(torch.cat([lens, lens_h]), 0, out=ptr[1:])
            return (ops.CsrCounts(ptr, torch.cat([c.indices, idx_h]), torch.cat([c.data, dat_h]), c.G),
                    torch.cat([f.double().contiguous(), plan_k.fetch(f.double().contiguous().reshape(-1, 1)).reshape(-1)]))
        self.cS, self.fS = with_halo(cS, fS)
        self.cU, self.fU = with_halo(cU, fU)
        # graph of the E rows in the row numbering of the local count CSR ([own | k_out])
        self.g_idx = ops.localize_rows(grow_e, self.c0, self.c1, self.k_out)           # (n_E, k + 1) int32
        self.g_w = wrow_e
        self.kp1 = self.k + 1
        self.corr = torch.empty((self.nloc, self.nrndm), dtype=self.dtype, device=dev)
        self.gamma = None
        self.stage_ms = np.zeros(4)
        self._resident = None
        self._plan = None
        self.peak_block_bytes = 0

    def _knn(self, pcs_full: torch.Tensor):
        """Exact kNN of the own cells among all cells: brute force up to PRUNE_FROM cells, projection-pruned beyond (the
        brute-force search is O(C^2): about 2 s per pass at 1M cells; the pruned one evaluated 10 % of the pairs there, 388 ms,
        and returns the same lists, ops.knn_search_pruned)."""
        if self.knn_mode == "pruned" or (self.knn_mode == "auto" and self.C >= PRUNE_FROM):
            return ops.knn_search_pruned(pcs_full, self.k, q0=self.c0, Q=self.nloc, stats=self.knn_stats)
        return ops.knn_search(pcs_full, self.k, include_self=False, q0=self.c0, Q=self.nloc)

    # ------------------------------------------------------------------ pooling of arbitrary E rows
    def _e_rows_of(self, g: torch.Tensor) -> torch.Tensor:
        """E-row numbers ([own | e_out]) of global cell numbers."""
        return ops.localize_rows(g, self.c0, self.c1, self.e_out).long()

    def _pool(self, counts: ops.CsrCounts, scale: torch.Tensor, erows, out: ops.CellMatrix) -> None:
        """out[i, :] = pooled vector of E row erows[i] (a slice = contiguous E rows, or an int64 tensor)."""
        if isinstance(erows, slice):
            gi, gw = self.g_idx[erows], self.g_w[erows]
        else:
            gi, gw = self.g_idx.index_select(0, erows), self.g_w.index_select(0, erows)
        n = int(gi.shape[0])
        if n == 0:
            return
        indptr = torch.arange(0, (n + 1) * self.kp1, self.kp1, device=self.dev, dtype=torch.int64)
        ops.knn_pool_csr(counts, scale, indptr, gi.reshape(-1), gw.reshape(-1), dtype=self.dtype, C_out=n, out=out, validate=False)

    def blocks(self) -> List[Tuple[int, int]]:
        return [(b, min(self.nloc, b + self.block_cells)) for b in range(0, self.nloc, self.block_cells)]

    def _plan_blocks(self) -> None:
        """Per block, once per graph: the e rows it reads outside itself (E-row numbers, ascending global number) and its
        neighbour lists in the numbering of its buffer [block | outside]; and the two buffers every block reuses."""
        self._plan = []
        single = len(self.blocks()) == 1
        max_rows, max_nb = 0, 0
        for (b0, b1) in self.blocks():
            nb_ix = self.neigh[b0:b1]
            if single:
                erows_out = torch.arange(self.nloc, self.nloc + self.n_e_halo, device=self.dev)
                ixs = ops.localize_rows(nb_ix, self.c0, self.c1, self.e_out)
            else:
                g = torch.unique(nb_ix.reshape(-1).long())                           # ascending global numbers
                outside = g[(g < self.c0 + b0) | (g >= self.c0 + b1)]
                erows_out = self._e_rows_of(outside)
                ixs = ops.localize_rows(nb_ix, self.c0 + b0, self.c0 + b1, outside)
            self._plan.append((b0, b1, erows_out, ixs))
            max_rows, max_nb = max(max_rows, (b1 - b0) + int(erows_out.numel())), max(max_nb, b1 - b0)
        self._ebuf = ops.CellMatrix.empty(max_rows, self.G, self.dtype)
        self._ubuf = ops.CellMatrix.empty(max_nb, self.G, self.dtype)
        self.peak_block_bytes = (self._ebuf.t.numel() + self._ubuf.t.numel()) * self._ebuf.t.element_size()

    # ------------------------------------------------------------------ one pass of the path
    def run(self, timed: bool = False) -> torch.Tensor:
        dev, G = self.dev, self.G
        if self._plan is None:
            self._plan_blocks()
        single = len(self._plan) == 1
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        tA = tB = tD = tK = 0.0
        # ---- A, first half: the exact kNN search of the own cells among all cells (analysis.py:1005).  The graph of a dataset
        #      does not change between passes - the count-row halo was built from it - but the search is part of
        #      knn_imputation and of the metric, so every pass repeats it
        ev[0].record()
        self._knn(self._pcs_full)
        ev[1].record()
        if timed:
            torch.cuda.synchronize()
            tK = ev[0].elapsed_time(ev[1])
        # ---- pass 1: A (pooling of the own cells) + B (fit_slope moments, estimation.py:267-279), block by block
        mom = torch.zeros((3, G), dtype=torch.float64, device=dev)
        abs_st = absmax = None
        for (b0, b1, erows_out, ixs) in self._plan:
            nb = b1 - b0
            ev[0].record()
            Sx_b, Ux_b = self._ebuf.rows(0, nb), self._ubuf.rows(0, nb)
            self._pool(self.cS, self.fS, slice(b0, b1), Sx_b)
            self._pool(self.cU, self.fU, slice(b0, b1), Ux_b)
            ev[1].record()
            mom += ops.fit_slope_moments(Ux_b, Sx_b)
            ev[2].record()
            if self.rules is None:                    # first pass only: the scale facts of e = Sx_sz over ALL cells of ALL ranks
                if self.dtype == torch.float64:       # the f64 sqrt element's domain: max |Sx| of every pooled block - the matrix itself, never
                    lo, hi = Sx_b.t.aminmax()         # the staging buffer (its halo rows are not written yet on the first pass); judged after
                    m = torch.maximum(lo.abs(), hi.abs()).double().reshape(1)     # the all-reduce below, by every rank alike
                    absmax = m if absmax is None else torch.maximum(absmax, m)
                st = ops.abs_stats(Sx_b)
                abs_st = st if abs_st is None else torch.stack([abs_st[0] + st[0], torch.minimum(abs_st[1], st[1]), abs_st[2] + st[2]])
            if timed:
                torch.cuda.synchronize()
                tA += ev[0].elapsed_time(ev[1]); tB += ev[1].elapsed_time(ev[2])
        if self.rules is None:
            # stage D's branch rule is decided ONCE from whole-matrix reductions, all-reduced: identical on every rank and for
            # every block size (a per-block or per-rank decision could differ on borderline data)
            if abs_st is None:                      # a rank without a block still takes part in the all-reduce: the neutral element
                abs_st = torch.tensor([0.0, float("inf"), 0.0], dtype=torch.float64, device=self._ebuf.t.device)
            stats = D.all_reduce_abs_stats(abs_st)
            if self.dtype == torch.float64:
                # a rank that raised on its own block would leave the others waiting in the collectives: the fact is reduced first (MAX),
                # then every rank takes the same decision
                if absmax is None:
                    absmax = torch.zeros(1, dtype=torch.float64, device=self._ebuf.t.device)
                m = float(D.all_reduce_max(absmax))
                if not m < ops.F64_SQRT_MAX:
                    raise ValueError(f"colDeltaCor sqrt transform in f64: |e| reaches {m:.3g}, outside the supported range (< {ops.F64_SQRT_MAX:g})")
            self.rules = ops.partial_rules_for(self._ebuf, ops.SQRT, self.psc, stats=stats, cells=self.C, domain_checked=True)
        ev[0].record()
        D.all_reduce_sum(mom)
        gamma = ops.fit_slope_from_moments(mom)
        gamma[~torch.isfinite(gamma)] = 0.0          # fit_gammas' policy for genes without signal (analysis.py:1260); NaN would poison every d[c]
        self.gamma = gamma
        ev[1].record()
        if timed:
            torch.cuda.synchronize()
            tB += ev[0].elapsed_time(ev[1])
        # ---- pass 2: C + D per block.  e rows = [block | sampled neighbours outside the block].  The blocks are walked
        #      backwards: the block pass 1 pooled last is still in the buffers and is not pooled again (one block = resident
        #      mode: nothing is pooled twice, only the halo rows are added)
        for n_done, (b0, b1, erows_out, ixs) in enumerate(reversed(self._plan)):
            nb, n_out = b1 - b0, int(erows_out.numel())
            ev[0].record()
            e_buf, Ux_b = self._ebuf.rows(0, nb + n_out), self._ubuf.rows(0, nb)
            if n_done > 0:
                self._pool(self.cS, self.fS, slice(b0, b1), e_buf.rows(0, nb))
                self._pool(self.cU, self.fU, slice(b0, b1), Ux_b)
            self._pool(self.cS, self.fS, erows_out, e_buf.rows(nb, nb + n_out))
            ev[1].record()
            ops.coldeltacor_partial_fused(e_buf, Ux_b, gamma, None, ixs, ops.SQRT, self.rules, self.psc, cell0=0, u_row0=0,
                                          out=self.corr[b0:b1], validate=False)
            ev[2].record()
            if timed:
                torch.cuda.synchronize()
                tA += ev[0].elapsed_time(ev[1]); tD += ev[1].elapsed_time(ev[2])
        if single:
            self._resident = (self._ebuf, self._ubuf)
        if timed:
            self.stage_ms += np.array([tA, tB, tK, tD])
        return self.corr

    def gathered_corr(self) -> torch.Tensor:
        """All cells' correlation rows on every rank (the RCCL all-gather north_star names)."""
        return D.all_gather_rows(self.corr, self.C)


def memory_plan(C: int, G: int, nnz_per_cell: float, world: int, block_cells: int, nrndm: int = 250, k: int = 30, count_bytes: int = 1,
                elem_bytes: int = 4, halo_e: float = 0.1, halo_k: float = 0.25) -> Dict[str, float]:
    """Bytes one rank holds (GB).  halo_e / halo_k: E and count-row halos as fractions of the rank's own cells (measured:
    tools/halo_fraction.py, AtlasPath.n_e_halo / n_count_halo)."""
    nloc = math.ceil(C / world)
    ld = ops.padded_ld(G)
    blk = min(block_cells if block_cells > 0 else nloc, nloc)
    csr = 2 * (nloc * (1 + halo_k)) * (nnz_per_cell * (4 + count_bytes) + 8 + 4 * (G // 2048 + 2))
    dense_block = (2 * blk + blk * (halo_e if blk == nloc else 0.6)) * ld * elem_bytes
    gb = lambda x: x / 1e9
    return {"cells_per_rank": nloc, "csr_layers_with_halo_GB": gb(csr), "pcs_embedding_replicated_GB": gb(C * 32 * 8),
            "graph_and_neighbour_lists_GB": gb(nloc * (1 + halo_e) * (k + 1) * 8 + nloc * nrndm * 4), "block_Sx_Ux_GB": gb(dense_block),
            "corr_rows_GB": gb(nloc * nrndm * elem_bytes), "kNN_workspace_GB": gb(8192 * C * 4),
            "total_GB": gb(csr + C * 32 * 8 + dense_block + nloc * nrndm * (4 + elem_bytes) + 8192 * C * 4)}


# ---------------------------------------------------------------------------------------------------------------------------
# synthetic atlas: the generator of bench.py's dense dataset in block-keyed form, so that any rank can produce
# exactly its own cells; counts are thinned to the requested density and stored CSR
def synth_atlas(C: int, G: int, P: int, dev, density: float = 0.08, c0: int = 0, c1: Optional[int] = None, seed: int = 20180812):
    """Returns (cS, cU, totS, totU, pcs_own, emb_own) for cells c0..c1-1 of a C-cell dataset whose labels already follow
    the Hilbert curve of the embedding.  Identical for every (c0, c1) split."""
    c1 = C if c1 is None else c1
    gen = torch.Generator(device=dev).manual_seed(seed)
    alpha = torch.exp(torch.randn(G, generator=gen, device=dev))
    gam = torch.exp(-0.5 + 0.5 * torch.randn(G, generator=gen, device=dev))
    t_on = torch.rand(G, generator=gen, device=dev) * 0.7
    switching = (torch.rand(G, generator=gen, device=dev) < 0.6).float()
    branch_gene = (torch.rand(G, generator=gen, device=dev) < 0.3).float()
    # latent state of ALL cells (small vectors), embedding, curve order
    t = torch.rand(C, generator=gen, device=dev)
    branch = (torch.rand(C, generator=gen, device=dev) < 0.5).float()
    size = torch.exp(0.3 * torch.randn(C, generator=gen, device=dev))
    wob = torch.randn((C, 2), generator=gen, device=dev, dtype=torch.float64)
    emb = torch.stack([t.double() * 10.0, (branch.double() * 2 - 1) * torch.clamp(t.double() - 0.5, min=0) * 8.0], 1) + 0.35 * wob
    perm = ops.hilbert_order(emb.contiguous()).long()
    t, branch, size, emb = t[perm], branch[perm], size[perm], emb[perm].contiguous()
    noise = torch.randn((C, P - 2), generator=gen, device=dev, dtype=torch.float64)[perm] * 0.05
    pcs = torch.cat([emb, noise], 1).contiguous()

    def rates(s, e):
        tt = t[s:e, None]
        tau = torch.clamp(tt - t_on[None, :], min=0.0) * switching[None, :] + (1 - switching[None, :]) * 1.0
        gate = 1.0 - branch_gene[None, :] * branch[s:e, None] * (tt > 0.5).float()
        u = alpha[None, :] * (1 - torch.exp(-4.0 * tau)) * gate
        sp = (alpha / gam)[None, :] * (1 - torch.exp(-2.0 * gam[None, :] * tau)) * gate
        return 0.3 * size[s:e, None] * u, size[s:e, None] * sp
    # thinning factor for the requested density of the spliced layer, calibrated on the first chunk (deterministic)
    _, lam = rates(0, min(C, KEY_CHUNK))
    lo, hi = 1e-6, 10.0
    for _ in range(40):
        mid = math.sqrt(lo * hi)
        if float((1 - torch.exp(-mid * lam)).mean()) > density:
            hi = mid
        else:
            lo = mid
    thin = math.sqrt(lo * hi)
    ld = ops.padded_ld(G)
    parts = {"S": [], "U": []}
    for ch in range(c0 // KEY_CHUNK, (c1 - 1) // KEY_CHUNK + 1):
        s, e = ch * KEY_CHUNK, min(C, (ch + 1) * KEY_CHUNK)
        g2 = torch.Generator(device=dev).manual_seed(seed * 7919 + ch)
        lu, ls = rates(s, e)
        cu = torch.poisson(thin * lu, generator=g2).clamp_(max=65535)
        cs = torch.poisson(thin * ls, generator=g2).clamp_(max=65535)
        a, b = max(c0, s) - s, min(c1, e) - s
        for name, m in (("S", cs), ("U", cu)):
            d = torch.zeros((b - a, ld), dtype=torch.int16, device=dev)
            d[:, :G] = m[a:b].to(torch.int32).to(torch.int16)
            parts[name].append(ops.CsrCounts.from_dense(ops.CountMatrix(d, G).narrowed()))

    def cat(ps):
        wide = any(p.data.dtype == torch.int16 for p in ps)
        ptr, base = [torch.zeros(1, dtype=torch.int64, device=dev)], 0
        for p in ps:
            ptr.append(p.indptr[1:] + base)
            base += p.nnz
        return ops.CsrCounts(torch.cat(ptr), torch.cat([p.indices for p in ps]),
                             torch.cat([(p.data.to(torch.int16) if wide else p.data) for p in ps]), G)
    cS, cU = cat(parts["S"]), cat(parts["U"])
    return cS, cU, cS.row_sums(), cU.row_sums(), pcs[c0:c1].contiguous(), emb[c0:c1].contiguous()


def size_factors(totS: torch.Tensor, totU: torch.Tensor, C: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """avg_size / cell_size of _normalize_S / _normalize_U (analysis.py:540-549, 570-579) with the mean over ALL cells."""
    sums = torch.stack([totS.sum(), totU.sum()])
    D.all_reduce_sum(sums)
    return (sums[0] / C) / totS.clamp(min=1.0), (sums[1] / C) / totU.clamp(min=1.0)


def auto_block_cells(nloc: int, C: int, G: int, dev, elem_bytes: int = 4) -> int:
    """Cells per streamed block chosen from the free HBM: as many as fit beside the CSR layers - a block holds ~2.6 dense
    rows per cell (Sx + the e rows it reads outside itself, Ux) of `elem_bytes` per element; the kNN workspace of 8192 queries x C
    distances and 6 GB of headroom stay free.  On 288 GB: 1M cells x 30k genes walk in three blocks of ~366 000 cells in f32."""
    free = torch.cuda.mem_get_info(dev)[0] - 8192 * C * 4 - (6 << 30)
    return int(max(4096, min(nloc, free * 0.6 // (2.6 * ops.padded_ld(G) * elem_bytes))))


def bench_main(a, dev, rank: int, world: int) -> Optional[dict]:
    """bench.py --workload cfg5: the streamed path on synthetic CSR layers; returns the JSON record on rank 0."""
    C, G = a.cells, a.genes
    c0, c1 = D.shard_bounds(C, world, rank)
    t0 = time.perf_counter()
    cS, cU, totS, totU, pcs, emb = synth_atlas(C, G, a.pca_dims, dev, density=a.density, c0=c0, c1=c1)
    fS, fU = size_factors(totS, totU, C)
    nloc = c1 - c0
    dt_name = getattr(a, "dtype", "f32")                 # bench.py --dtype: f64 = the reference's arithmetic (the default), f32 = production mode
    tdt = torch.float64 if dt_name == "f64" else torch.float32
    block = a.block_cells if a.block_cells > 0 else auto_block_cells(nloc, C, G, dev, 8 if dt_name == "f64" else 4)
    path = AtlasPath(cS, cU, fS, fU, pcs, emb, c0=c0, C_total=C, k=a.k, n_neighbors=a.n_neighbors, sampled_fraction=a.sampled_fraction,
                     block_cells=block, dtype=tdt, knn=getattr(a, "knn", "auto"))
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t0
    for _ in range(a.warmup):
        path.run()

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        path.run(timed=True)
        if world > 1:
            path.gathered_corr()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if a.dump:
        corr_all = path.gathered_corr() if world > 1 else path.corr
        neigh_all = D.all_gather_rows(path.neigh, C)
        if rank == 0:
            np.savez(a.dump, gamma=path.gamma.cpu().numpy(), corr=corr_all.cpu().numpy(), neigh=neigh_all.cpu().numpy())
    if rank != 0:
        return None
    ms = dt / a.steps * 1e3
    st = path.stage_ms / a.steps
    nnz = cS.nnz / max(1, nloc)
    plan_1m = memory_plan(1_000_000, G, nnz, 8, 0, nrndm=path.nrndm, k=a.k, count_bytes=cS.data.element_size(),
                          elem_bytes=8 if dt_name == "f64" else 4, halo_e=path.n_e_halo / nloc, halo_k=path.n_count_halo / nloc)
    return {
        "metric": f"cells/sec through knn_imputation->fit_slope->colDeltaCor, {C} cells x {G} genes (cfg5: CSR layers, streamed blocks)",
        "value": C / (ms * 1e-3), "unit": "cells/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": dt_name, "data": "synthetic",
        "rccl_ranks": dist.get_world_size() if (dist.is_initialized() and dist.get_backend() == "nccl") else 0,
        "config": {"workload": f"cfg5 (BASELINE.json configs[4], scaled to what one run holds): synthetic {C} cells x {G} genes, CSR count layers at "
                               f"{100.0 * cS.nnz / max(1, nloc) / G:.1f} % density ({nnz:.0f} non-zeros per cell, {'uint8' if cS.data.dtype == torch.uint8 else 'uint16'} counts), "
                               f"streamed in blocks of {path.block_cells} cells: knn_imputation(k={a.k}) from CSR -> fit_slope -> velocity chain -> "
                               f"colDeltaCorSqrtpartial(nrndm={path.nrndm})",
                   "cells": C, "genes": G, "k": a.k, "nrndm": path.nrndm, "blocks_per_rank": len(path.blocks()), "block_cells": path.block_cells,
                   "e_sharded": True, "count_row_halo": path.n_count_halo, "e_halo_rows": path.n_e_halo,
                   "knn_search": ({"mode": "exact, projection-pruned", **path.knn_stats} if path.knn_stats else {"mode": "exact, brute force"}),
                   "stage_D_rule": ops.RULE_NAMES.get(path.rules, str(path.rules)),
                   "stage_ms": {"A_knn_search": st[2], "A_pooling_from_csr (own cells + e rows outside the block)": st[0], "B_fit_slope": st[1],
                                "D_coldeltacor": st[3]},
                   "setup_s": setup_s,
                   "resident_bytes": {"csr_layers_with_halo": path.cS.nbytes + path.cU.nbytes, "peak_block_Sx_Ux": path.peak_block_bytes,
                                      "torch_peak_allocated": int(torch.cuda.max_memory_allocated(dev))},
                   "plan_1M_cells_8_ranks_GB": plan_1m},
    }
