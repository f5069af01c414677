"""GPU tests at BASELINE.json's FULL size (50 000 cells x 30 000 genes, k = 30, nrndm = 250).

The oracle cannot run the whole problem in seconds, so parity at this size is established through
  (1) size-independent properties: linearity of the pooling operator, sortedness of neighbour lists,
      |r| <= 1, exact antisymmetry corr(e, -d) = -corr(e, d), scale invariance corr(e, a*d) = corr(e, d),
      independence from the scheduling order;
  (2) spot checks: randomly sampled queries / cells / genes recomputed by the fp64 CPU oracle from the
      rows they touch (downloaded from the device), compared at the f32 tolerances of DESIGN.md.
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C, G, K, NN, FRAC = 50000, 30000, 30, 500, 0.5


@pytest.fixture(scope="module")
def world():
    sys.path.insert(0, ROOT)
    import velocyto_amd
    from velocyto_amd import ops
    import bench
    dev = ops.require_gpu()
    if torch.cuda.get_device_properties(0).total_memory < 100e9:
        pytest.skip("full-size test needs a >= 100 GB device")
    S, U, pcs = bench.synth(C, G, 30, dev)
    neigh, _ = bench.sample_neighbors_device(pcs[:, :2].contiguous(), NN, FRAC, dev)
    return dict(ops=ops, dev=dev, S=S, U=U, pcs=pcs, neigh=neigh)


def bench_counts():
    import bench
    from velocyto_amd import ops
    return bench.synth_counts(C, G, 30, ops.require_gpu())


def _pool_inputs(w, ops):
    idx, dist = ops.knn_search(w["pcs"], K)
    conn = (dist > 0).float()
    wrow = torch.cat([torch.ones((C, 1), device=w["dev"]), conn], 1)
    wrow = (wrow / wrow.sum(1, keepdim=True)).contiguous()
    indices = torch.cat([torch.arange(C, device=w["dev"], dtype=torch.int32)[:, None], idx], 1).contiguous()
    indptr = torch.arange(0, (C + 1) * (K + 1), K + 1, device=w["dev"], dtype=torch.int64)
    return idx, dist, indptr, indices, wrow


def test_fullsize_pipeline_properties_and_spot_checks(world, oracle):
    w, ops = world, world["ops"]
    rng = np.random.default_rng(11)
    S, U, pcs, neigh = w["S"], w["U"], w["pcs"], w["neigh"]
    # ---------------- A: kNN
    idx, dist, indptr, indices, wrow = _pool_inputs(w, ops)
    assert bool((dist[:, 1:] >= dist[:, :-1]).all()), "neighbour lists must be sorted by distance"
    assert bool((idx != torch.arange(C, device=w["dev"], dtype=torch.int32)[:, None]).all()), "query must be excluded"
    qs = rng.choice(C, 24, replace=False)
    P = pcs.cpu().numpy()
    for q in qs:
        d2 = ((P - P[q]) ** 2).sum(1)
        d2[q] = np.inf
        ref = np.lexsort((np.arange(C), d2))[:K]
        assert np.array_equal(idx[q].cpu().numpy(), ref)
        np.testing.assert_allclose(dist[q].cpu().numpy(), np.sqrt(d2[ref]), rtol=1e-12)
    # ---------------- A: pooling -- linearity and spot check
    Sx = ops.knn_pool(S, indptr, indices, wrow, validate=False)
    Ux = ops.knn_pool(U, indptr, indices, wrow, validate=False)
    SU = ops.CellMatrix(S.t * 2.0 + U.t, G)
    both = ops.knn_pool(SU, indptr, indices, wrow, validate=False, order=ops.morton_order(pcs, 3), slab_genes=1024)
    err = (both.t - (2.0 * Sx.t + Ux.t)).abs().max().item()
    scale = both.t.abs().max().item()
    assert err <= 4e-6 * scale, (err, scale)
    del SU, both
    cells = rng.choice(C, 8, replace=False)
    for c in cells:
        rows = S.t[indices[c].long(), :G].double().cpu().numpy()
        ref = (wrow[c].double().cpu().numpy()[:, None] * rows).sum(0)
        np.testing.assert_allclose(Sx.t[c, :G].cpu().numpy(), ref, rtol=3e-6, atol=1e-6)
    # pooling from the integer count layers (uint8 / uint16 storage) == pooling of the float matrices they came from
    cS, cU, fS, fU, _ = bench_counts()
    for narrow in (True, False):
        a8, b8 = (cS, cU) if narrow else (ops.CountMatrix(cS.t.to(torch.int16), G), ops.CountMatrix(cU.t.to(torch.int16), G))
        Sc, Uc = ops.knn_pool_counts(a8, b8, fS, fU, indptr, indices, wrow, dtype=torch.float32, validate=False, order=ops.morton_order(pcs, 3))
        assert (Sc.t - Sx.t).abs().max().item() <= 4e-6 * Sx.t.abs().max().item()
        assert (Uc.t - Ux.t).abs().max().item() <= 4e-6 * Ux.t.abs().max().item()
        del Sc, Uc
    del cS, cU
    # ---------------- B: fit_slope spot check on sampled genes
    gam = ops.fit_slope(Ux, Sx)
    genes = rng.choice(G, 48, replace=False)
    xs, ys = Sx.t[:, genes].double().cpu().numpy(), Ux.t[:, genes].double().cpu().numpy()
    ref = np.maximum(0, (xs * ys).sum(0) / (xs * xs).sum(0))
    got = gam[genes].cpu().numpy()
    ok = np.isfinite(ref)
    np.testing.assert_allclose(got[ok], ref[ok], rtol=2e-5, atol=1e-7)
    # ---------------- C: velocity chain spot check
    out = ops.velocity_chain(Sx, Ux, gam, None, want=("delta_S", "dmat"), transform=ops.SQRT, psc=1e-10)
    dmat = out["dmat"]
    g32 = gam.cpu().numpy()
    for c in cells[:4]:
        s, u = Sx.t[c, :G].double().cpu().numpy(), Ux.t[c, :G].double().cpu().numpy()
        vel = u - g32.astype(np.float64) * s
        D = (s + vel) - s
        ref = np.sign(D) * np.sqrt(np.abs(D) + 1e-10)
        okc = np.isfinite(ref)
        np.testing.assert_allclose(dmat.t[c, :G].double().cpu().numpy()[okc], ref[okc], rtol=1e-4, atol=2e-3)
    del out
    # ---------------- D: correlation properties
    corr = ops.coldeltacor_partial(Sx, dmat, neigh, ops.SQRT, ops.RULES_PARTIAL, 1e-10, validate=True)
    fin = torch.isfinite(corr)
    assert fin.float().mean().item() > 0.999
    assert corr[fin].abs().max().item() <= 1 + 1e-5
    neg = ops.CellMatrix(-dmat.t, G)
    c_neg = ops.coldeltacor_partial(Sx, neg, neigh, ops.SQRT, ops.RULES_PARTIAL, 1e-10, validate=False)
    assert torch.equal(torch.isfinite(c_neg), fin) and torch.equal(c_neg[fin], -corr[fin]), "corr(e, -d) must be exactly -corr(e, d)"
    neg.t.mul_(-3.0)                                    # now 3 * dmat
    c_scaled = ops.coldeltacor_partial(Sx, neg, neigh, ops.SQRT, ops.RULES_PARTIAL, 1e-10, validate=False,
                                       order=ops.morton_order(pcs[:, :2], 2))
    assert (c_scaled[fin] - corr[fin]).abs().max().item() < 2e-5
    del neg, c_neg, c_scaled
    # velocity chain folded into the kernel == materialised dmat, bit for bit, main part and last-round tiles alike
    fused = ops.coldeltacor_partial_fused(Sx, Ux, gam, None, neigh, ops.SQRT, ops.RULES_PARTIAL, 1e-10, validate=False)
    assert torch.equal(torch.nan_to_num(fused, nan=7.0), torch.nan_to_num(corr, nan=7.0))
    del fused
    # ---------------- D: spot check against the fp64 oracle on the rows the sampled cells touch
    # (cells C-5 and C-700 sit in the groups beyond the last full round, which run as narrow column tiles)
    for c in list(cells[:3]) + [C - 5, C - 700]:
        nb = neigh[c].long().cpu().numpy()
        rows = np.concatenate([[c], nb])
        e_sub = Sx.t[torch.as_tensor(rows, device=w["dev"]), :G].double().cpu().numpy().T      # (G, 1 + nrndm)
        d_sub = np.zeros_like(e_sub)
        d_sub[:, 0] = dmat.t[c, :G].double().cpu().numpy()
        ixs = np.zeros((e_sub.shape[1], len(nb)), dtype=np.int64)
        ixs[0] = np.arange(1, len(nb) + 1)
        ref = oracle.coldeltacor_partial_compact(e_sub, d_sub, ixs, "sqrt", 1e-10, c0=0, c1=1)[0]
        got = corr[c].cpu().numpy()
        okc = np.isfinite(ref)
        np.testing.assert_allclose(got[okc], ref[okc], atol=5e-5)


def test_fullsize_markov_chain_factored_vs_dense(world):
    """prepare_markov / run_markov at 50 000 cells: the factored chain (no (n, n) matrix) against the dense matrix it stands for
    (10 GB in f32, streamed by k_vecmat_dense_vec), plus what must hold at any size: every iterate is a probability vector
    (tr is row-stochastic), time evolution and path integral agree with each other, and the result does not depend on whether the
    steps are replayed from a hipGraph."""
    ops, dev = world["ops"], world["dev"]
    emb = world["pcs"][:, :2].double().contiguous()
    neigh = world["neigh"].to(torch.int64)
    n, m = neigh.shape
    gen = torch.Generator(device=dev).manual_seed(7)
    tp = torch.rand((n, m), generator=gen, device=dev, dtype=torch.float64) + 0.05
    tp /= tp.sum(1, keepdim=True)
    indptr = torch.arange(0, n * m + 1, m, device=dev)
    sd = float(emb.std()) * 0.05
    fac = ops.prepare_markov_factored(indptr, neigh.ravel(), tp.ravel(), emb, sd, 2 * sd, compute_dtype=torch.float32)
    x0 = torch.rand(n, generator=gen, device=dev, dtype=torch.float64)
    x0 /= x0.sum()
    x5, _ = ops.diffuse(x0, fac, 5, accumulate=False)
    assert abs(float(x5.sum()) - 1.0) < 5e-6 and float(x5.min()) >= 0.0
    x40, acc40 = ops.diffuse(x0, fac, 40, accumulate=True)                  # graph-replayed
    x35, _ = ops.diffuse(x5, fac, 35, accumulate=False)                     # 5 + 35 eager/graph mix
    assert torch.allclose(x40, x35, rtol=1e-12, atol=0)
    assert abs(float(acc40.sum()) - 40.0) < 2e-4
    dense = fac.dense(torch.float32)                                         # the (n, n) matrix the reference builds
    assert dense.shape == (n, n)
    rows = dense[:: n // 64].double().sum(1)
    assert float((rows - 1).abs().max()) < 1e-5
    d5, _ = ops.diffuse(x0, dense, 5, accumulate=False)
    rel = ((x5 - d5).abs() / d5.clamp_min(1e-300)).max()
    assert float(rel) < 5e-5, float(rel)
    del dense
