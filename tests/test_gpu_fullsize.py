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
    # ---------------- D: correlation properties (the rule the callers use on this matrix: f32 -> the no-pseudocount form)
    rules = ops.partial_rules_for(Sx, ops.SQRT, 1e-10)
    assert rules == ops.RULES_PARTIAL_NOPSC
    corr = ops.coldeltacor_partial(Sx, dmat, neigh, ops.SQRT, rules, 1e-10, validate=True)
    fin = torch.isfinite(corr)
    assert fin.float().mean().item() > 0.999
    assert corr[fin].abs().max().item() <= 1 + 1e-5
    neg = ops.CellMatrix(-dmat.t, G)
    c_neg = ops.coldeltacor_partial(Sx, neg, neigh, ops.SQRT, rules, 1e-10, validate=False)
    assert torch.equal(torch.isfinite(c_neg), fin) and torch.equal(c_neg[fin], -corr[fin]), "corr(e, -d) must be exactly -corr(e, d)"
    neg.t.mul_(-3.0)                                    # now 3 * dmat
    c_scaled = ops.coldeltacor_partial(Sx, neg, neigh, ops.SQRT, rules, 1e-10, validate=False,
                                       order=ops.morton_order(pcs[:, :2], 2))
    assert (c_scaled[fin] - corr[fin]).abs().max().item() < 2e-5
    del neg, c_neg, c_scaled
    # velocity chain folded into the kernel == materialised dmat, bit for bit, main part and last-round tiles alike
    fused = ops.coldeltacor_partial_fused(Sx, Ux, gam, None, neigh, ops.SQRT, rules, 1e-10, validate=False)
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


def test_fullsize_stage_d_256_cells_against_the_oracle(world, oracle):
    """Stage D at the headline size against the fp64 oracle on 256 whole cells = 64 000 pairs x 30 000 genes: 192 cells spread
    over the dataset + 64 from the groups beyond the last full round of the launch (they run as narrower column tiles: every
    tile of such a cell is covered, all 250 columns are compared).  f32 production rule and the literal rule, fused launch."""
    w, ops = world, world["ops"]
    dev = w["dev"]
    rng = np.random.default_rng(23)
    S, U, pcs, neigh = w["S"], w["U"], w["pcs"], w["neigh"]
    idx, dist, indptr, indices, wrow = _pool_inputs(w, ops)
    indices, wrow = ops.canonical_graph_rows(indices.reshape(C, K + 1), wrow)
    Sx = ops.knn_pool(S, indptr, indices, wrow, validate=False)
    Ux = ops.knn_pool(U, indptr, indices, wrow, validate=False)
    gam = ops.fit_slope(Ux, Sx)
    gam[~torch.isfinite(gam)] = 0.0
    g64 = gam.double().cpu().numpy()
    tail0 = (C // 8 // 256) * 256 * 8                      # first cell of the tiled tail part (natural schedule, 8-cell groups, 256 CUs)
    assert 0 < C - tail0 < 2048
    cells = np.concatenate([rng.choice(tail0, 192, replace=False), rng.choice(np.arange(tail0, C), 64, replace=False)])
    got = {}
    for name, rules in (("production", ops.partial_rules_for(Sx, ops.SQRT, 1e-10)), ("literal", ops.RULES_PARTIAL)):
        got[name] = ops.coldeltacor_partial_fused(Sx, Ux, gam, None, neigh, ops.SQRT, rules, 1e-10, validate=False)[torch.as_tensor(cells, device=dev)].cpu().numpy()
    worst = {"production": 0.0, "literal": 0.0}
    B = 64                                                 # cells per oracle call (its OpenMP loop runs over cells): <= 16 064 rows of e on the host
    for b in range(0, len(cells), B):
        cs = cells[b:b + B]
        nb = neigh[torch.as_tensor(cs, device=dev)].long().cpu().numpy()
        others = np.setdiff1d(np.unique(nb.ravel()), cs)
        rows = np.concatenate([cs, others])                 # the batch's cells first: columns 0 .. B-1 of the sub-problem
        col = np.full(C, -1, dtype=np.int64)
        col[rows] = np.arange(len(rows))
        e_sub = Sx.t[torch.as_tensor(rows, device=dev), :G].double().cpu().numpy().T.copy()    # (G, rows), the oracle's layout
        s, u = e_sub[:, :len(cs)], Ux.t[torch.as_tensor(cs, device=dev), :G].double().cpu().numpy().T
        Dv = (s + (u - g64[:, None] * s)) - s
        d_sub = np.zeros_like(e_sub)
        d_sub[:, :len(cs)] = np.sign(Dv) * np.sqrt(np.abs(Dv) + 1e-10)
        ixs = np.zeros((len(rows), nb.shape[1]), dtype=np.int64)
        ixs[:len(cs)] = col[nb]
        ref = oracle.coldeltacor_partial_compact(e_sub, d_sub, ixs, "sqrt", 1e-10, c0=0, c1=len(cs))[:len(cs)]
        ok = np.isfinite(ref)
        for name in got:
            g_ = got[name][b:b + B]
            assert np.array_equal(np.isnan(g_), ~ok)
            worst[name] = max(worst[name], float(np.abs(g_[ok] - ref[ok]).max()))
        del e_sub, d_sub
    assert worst["production"] <= 5e-5 and worst["literal"] <= 5e-5, worst


def test_default_sight_balanced_knn_at_50k_cells(world):
    """knn_imputation(balanced=True) with the reference's DEFAULT sight at the headline size (analysis.py:985-988: b_sight = b_maxl =
    N - 1, i.e. (50 000 x 50 000) sight lists): the lists stay on the device (10 GB int32 + 20 GB fp64), the greedy loop reads an
    int32 host copy (vcy_balance_knn_host32), 4 s in all - int64 + fp64 host lists would be 40 GB.  With maxl = N - 1 the in-degree cap
    can never bind, so the balanced graph must BE the plain 30-NN graph (a size-independent property), self in column 0."""
    import velocyto_amd
    w, ops = world, world["ops"]
    P = w["pcs"].cpu().numpy()
    bk = velocyto_amd.neighbors.BalancedKNN(k=K, sight_k=C - 1, maxl=C - 1, n_jobs=4).fit(P)
    d_new, dsi_new, l = bk.kneighbors()
    idx, dist = ops.knn_search(P, K)
    assert dsi_new.shape == (C, K + 1) and np.array_equal(dsi_new[:, 0], np.arange(C))
    assert np.array_equal(dsi_new[:, 1:], idx.cpu().numpy())
    np.testing.assert_array_equal(d_new[:, 1:], dist.cpu().numpy())
    assert np.array_equal(l, np.bincount(idx.cpu().numpy().ravel(), minlength=C)) and l.max() < C - 1
    del bk


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


def test_fullsize_f32_against_f64_all_pairs(world):
    """The production arithmetic (f32 storage, f32 lane accumulators) against the reference's (fp64 throughout,
    speedboosted.pyx:13-538) on the SAME inputs at 50 000 x 30 000: every pooled value, every gamma and ALL 12.5 M
    correlations, not a sample.  Stated tolerances: pooled matrices 2e-6 relative to the matrix scale, gammas 2e-6
    relative, correlations 5e-5 absolute (the f32 bar of test_gpu_ops.py; measured 2e-6)."""
    w, ops = world, world["ops"]
    dev = w["dev"]
    cS, cU, fS, fU, pcs = bench_counts()
    idx, dist = ops.knn_search(pcs, K)
    conn = (dist > 0).double()
    wrow = torch.cat([torch.ones((C, 1), device=dev, dtype=torch.float64), conn], 1)
    wrow = (wrow / wrow.sum(1, keepdim=True)).contiguous()
    indices = torch.cat([torch.arange(C, device=dev, dtype=torch.int32)[:, None], idx], 1).contiguous()
    indptr = torch.arange(0, (C + 1) * (K + 1), K + 1, device=dev, dtype=torch.int64)
    neigh, _ = __import__("bench").sample_neighbors_device(pcs[:, :2].contiguous(), NN, FRAC, dev)
    res = {}
    for dt in (torch.float32, torch.float64):
        Sx, Ux = ops.knn_pool_counts(cS, cU, fS, fU, indptr, indices, wrow.to(dt), dtype=dt, validate=False)
        gam = ops.fit_slope(Ux, Sx)
        rules = ops.partial_rules_for(Sx, ops.SQRT, 1e-10)      # what the facade and the bench pass: f32 drops the pseudocount, f64 is literal
        assert rules == (ops.RULES_PARTIAL_NOPSC if dt == torch.float32 else ops.RULES_PARTIAL)
        corr = ops.coldeltacor_partial_fused(Sx, Ux, gam, None, neigh, ops.SQRT, rules, 1e-10, validate=False,
                                             order=ops.hilbert_order(pcs[:, :2].contiguous()))
        if dt == torch.float32:
            res[dt] = (Sx.t.clone(), Ux.t.clone(), gam.clone(), corr.clone())
        else:
            s32, u32, g32, c32 = res[torch.float32]
            for name, a32, a64 in (("Sx", s32, Sx.t), ("Ux", u32, Ux.t)):
                scale = float(a64.abs().max())
                worst = max(float((a32[r0:r0 + 5000].double() - a64[r0:r0 + 5000]).abs().max()) for r0 in range(0, C, 5000))
                assert worst <= 2e-6 * scale, (name, worst, scale)
            pos = gam > 0
            assert float(((g32.double() - gam.double()).abs() / gam.double().abs().clamp(min=1e-30))[pos].max()) <= 2e-6
            assert torch.equal(torch.isnan(c32), torch.isnan(corr)), "NaN pattern (zero-variance pairs) must not depend on the storage type"
            fin = torch.isfinite(corr)
            dmax = float((c32[fin].double() - corr[fin]).abs().max())
            assert int(fin.sum()) > 0.999 * C * neigh.shape[1] and dmax <= 5e-5, dmax
            print(f"f32 vs f64 over {int(fin.sum())} pairs: max |dcorr| = {dmax:.3e}")
        del Sx, Ux


def _load_make_golden():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)            # top level imports only numpy / stdlib; the reference is touched by load_reference() alone
    return mod


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_cfg1_fit_gammas_defaults_through_a_loom_file(tmp_path, golden, oracle, dtype):
    """BASELINE.json configs[0] at its stated size: 3000 cells x 2000 genes written to a .loom file, read back
    by VelocytoLoom(path), normalize -> knn_imputation(k=30) -> fit_gammas() with ALL defaults, against what the reference
    returned on the same arrays (tests/golden/cfg1.npz holds only its gammas / q / R2 and two checksums).  Tolerance of
    SURVEY section 7: parameters agree to rtol 1e-4 on >= 99 % of the genes, and where they do not the exact solution's
    objective is never worse than the reference's L-BFGS-B stopping point."""
    import velocyto_amd
    from velocyto_amd import loom_io
    g = golden("cfg1")
    mg = _load_make_golden()
    assert int(g["seed"]) == mg.CFG1_SEED
    S, U, pcs = mg.cfg1_inputs()
    path = str(tmp_path / "cfg1.loom")
    loom_io.write_loom(path, {"spliced": S, "unspliced": U, "ambiguous": np.zeros_like(S)},
                       {"CellID": np.array([f"c{i}" for i in range(S.shape[1])])}, {"Gene": np.array([f"g{i}" for i in range(S.shape[0])])})
    vlm = velocyto_amd.analysis.VelocytoLoom(path, dtype=dtype)
    assert np.array_equal(vlm.S, S) and np.array_equal(vlm.U, U)
    vlm.normalize("both", size=True, log=True)
    vlm.pcs = pcs
    vlm.knn_imputation(k=mg.CFG1_K, n_pca_dims=mg.CFG1_P, n_jobs=4)
    assert np.array_equal(np.sort(vlm.knn[0].indices), g["knn_row0"])
    rt = 1e-11 if dtype == "float64" else 3e-6
    np.testing.assert_allclose(vlm.Sx_sz[17], g["Sx_row17"], rtol=rt, atol=rt)
    np.testing.assert_allclose([vlm.Sx_sz.sum(), vlm.Ux_sz.sum()], [float(g["Sx_sum"]), float(g["Ux_sum"])], rtol=1e-6)
    vlm.fit_gammas()
    assert vlm.gammas.dtype == np.float32 and vlm.gammas.shape == (S.shape[0],)
    rel = lambda a, b: np.abs(a - b) / np.maximum(np.abs(b), 1e-3)
    rg, rq = rel(vlm.gammas.astype(float), g["gammas"].astype(float)), rel(vlm.q.astype(float), g["q"].astype(float))
    assert np.mean(rg > 1e-4) <= 0.01, (np.mean(rg > 1e-4), rg.max())
    assert np.mean(rq > 1e-3) <= 0.02, (np.mean(rq > 1e-3), rq.max())
    same = (rg <= 1e-4) & (rq <= 1e-3)                       # R2 is a function of the fitted parameters: compare where they agree;
    np.testing.assert_allclose(vlm.R2[same], g["R2"][same], atol=2e-3)     # the other genes are judged by the objective below
    # objective of the default weighted fit at both solutions, in fp64 on the facade's own pooled matrices
    X, Y = np.asarray(vlm.Sx_sz, dtype=np.float64), np.asarray(vlm.Ux_sz, dtype=np.float64)
    W = oracle.gamma_weights(np.asarray(vlm.Sx, dtype=np.float64), np.asarray(vlm.Ux, dtype=np.float64), X, Y, "maxmin_diag")
    f = lambda m, q: np.sum(W * (-Y + X * m[:, None] + q[:, None]) ** 2, 1)
    ours, ref = f(vlm.gammas.astype(float), vlm.q.astype(float)), f(g["gammas"].astype(float), g["q"].astype(float))
    slack = 1e-4 if dtype == "float64" else 2e-3
    assert np.all(ours <= ref * (1 + slack) + 1e-6), float(np.max(ours - ref))


def test_cfg2_size_balanced_knn_imputation_and_fit_slope(oracle):
    """BASELINE.json configs[1] at its stated size: 10 000 cells x 20 000 genes, k = 30, 30 PCs; the primary
    unbalanced graph and the secondary balanced=True, b_sight=240, b_maxl=120 (analysis.py:985-1003).  Properties of the whole
    result + spot checks against the oracle on the rows they touch."""
    import velocyto_amd
    from velocyto_amd import ops
    import bench
    dev = ops.require_gpu()
    Cc, Gg, k = 10_000, 20_000, 30
    cS, cU, fS, fU, pcs = bench.synth_counts(Cc, Gg, 30, dev, seed=20180810)
    vlm = velocyto_amd.analysis.VelocytoLoom.from_arrays(cS, cU, dtype="float32")
    vlm.normalize("both", size=True, log=False)
    vlm.pcs = pcs.cpu().numpy()
    rng = np.random.default_rng(5)
    P = vlm.pcs[:, :30]
    for balanced in (False, True):
        vlm.knn_imputation(k=k, n_pca_dims=30, balanced=balanced, b_sight=240, b_maxl=120, n_jobs=4)
        knn = vlm.knn.tocsr()
        deg_out = np.diff(knn.indptr)
        assert (deg_out == (k + 1 if balanced else k)).all()                     # balanced graph stores the self edge (distance 0)
        indeg = np.bincount(knn.indices, minlength=Cc) - (1 if balanced else 0)
        if balanced:
            assert indeg.max() <= 120, indeg.max()                               # the point of balancing: in-degree capped at b_maxl
        w = vlm.knn_smoothing_w.tocsr()
        assert np.allclose(np.asarray(w.sum(1)).ravel(), 1)
        # neighbours are the nearest within the sight radius: every listed neighbour is among the 240 nearest of its cell
        for c in rng.choice(Cc, 12, replace=False):
            d2 = ((P - P[c]) ** 2).sum(1)
            nearest = np.argsort(d2, kind="stable")[:241]
            nb = knn[c].indices
            assert np.isin(nb, nearest).all()
            if not balanced:
                assert set(nb) == set(nearest[1:k + 1])
            # pooled row == weights x size-normalised rows (fp64 on the host)
            cols = w[c].indices
            rows = np.stack([vlm.S_sz[:, j] for j in cols]).astype(np.float64)
            np.testing.assert_allclose(vlm.Sx[:, c], (w[c].data[:, None] * rows).sum(0), rtol=3e-6, atol=1e-6)
        vlm.fit_gammas(fit_offset=False, weighted=False)                          # fit_slope (estimation.py:267-279)
        genes = rng.choice(Gg, 64, replace=False)
        X, Y = vlm.dev("Sx_sz").t[:, genes].double().cpu().numpy(), vlm.dev("Ux_sz").t[:, genes].double().cpu().numpy()
        ref = np.maximum(0, (X * Y).sum(0) / (X * X).sum(0))
        ok = np.isfinite(ref)
        np.testing.assert_allclose(vlm.gammas[genes][ok], ref[ok], rtol=2e-5, atol=1e-7)
        # ---- the same step by the ORACLE on the whole 10 000 x 20 000 problem (fp64 restatement of analysis.py:982-1023):
        #      the graph bit for bit, every pooled value and every gamma at the f32 tolerances
        S_sz, U_sz = np.asarray(vlm.S_sz, dtype=np.float64), np.asarray(vlm.U_sz, dtype=np.float64)
        o_knn, o_w, o_Sx, o_Ux = oracle.knn_imputation(S_sz, U_sz, P, k=k, balanced=balanced, b_sight=240, b_maxl=120)
        if balanced:
            _, o_dist, o_dsi, o_l = oracle.balanced_knn_graph(P, k, 240, 120)
            bk = velocyto_amd.neighbors.BalancedKNN(k=k, sight_k=240, maxl=120, n_jobs=4).fit(P)
            d_new, dsi_new, l_new = bk.kneighbors()
            assert np.array_equal(dsi_new, o_dsi) and np.array_equal(l_new, o_l), "balanced graph differs from the oracle's"
            np.testing.assert_allclose(d_new, o_dist, rtol=1e-12, atol=1e-12)
            assert bk.dsi.shape == (Cc, 241) and bk.dist.shape == (Cc, 241)     # the sight lists stayed on the device until somebody asked
        o_csr = o_knn.tocsr(); o_csr.sort_indices()
        k_csr = knn.copy(); k_csr.sort_indices()
        assert np.array_equal(k_csr.indptr, o_csr.indptr) and np.array_equal(k_csr.indices, o_csr.indices), "kNN graph differs from the oracle's"
        np.testing.assert_allclose(k_csr.data, o_csr.data, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(np.asarray(vlm.Sx), o_Sx, rtol=3e-6, atol=1e-6)
        np.testing.assert_allclose(np.asarray(vlm.Ux), o_Ux, rtol=3e-6, atol=1e-6)
        o_g = oracle.fit_slope(o_Ux, o_Sx)
        okg = np.isfinite(o_g)
        np.testing.assert_allclose(vlm.gammas[okg], o_g[okg], rtol=2e-5, atol=1e-7)
        del S_sz, U_sz, o_Sx, o_Ux
    # ---- the reference's DEFAULT sight (analysis.py:985-988: b_sight = max(8k, N - 1) = the whole dataset, b_maxl = max(4k, N - 1)):
    #      (C, C) sight lists, kept on the device; graph against the oracle's, bit for bit
    vlm.knn_imputation(k=k, n_pca_dims=30, balanced=True, n_jobs=4)
    _, o_dist, o_dsi, o_l = oracle.balanced_knn_graph(P, k, Cc - 1, Cc - 1)
    got = vlm.knn.tocsr(); got.sort_indices()
    want = sparse_from_lists(o_dsi, o_dist, Cc)
    assert np.array_equal(got.indices, want.indices) and np.allclose(got.data, want.data, rtol=1e-12, atol=1e-12)
    bk = velocyto_amd.neighbors.BalancedKNN(k=k, sight_k=Cc - 1, maxl=Cc - 1, n_jobs=4).fit(P)
    d_new, dsi_new, l_new = bk.kneighbors()
    assert np.array_equal(dsi_new, o_dsi) and np.array_equal(l_new, o_l)


def sparse_from_lists(dsi, dist, n):
    from scipy import sparse
    m = sparse.csr_matrix((dist.ravel(), dsi.ravel(), np.arange(0, dsi.size + 1, dsi.shape[1])), shape=(n, n))
    m.sort_indices()
    return m


def test_fullsize_stage_d_reference_default_list_width_against_the_oracle(world, oracle):
    """Stage D at the headline size with the reference's DEFAULT neighbour lists (analysis.py:1452-1457, 1528-1572: n_neighbors =
    cells / 5 = 10 000, sampled_fraction = 0.3 => nrndm = 3000, walked in 12+ column tiles per group) against the fp64 oracle on
    128 whole cells x ALL 3000 columns x 30 000 genes (384 000 correlations per mode): six batches of 16 cells from across the launch
    schedule and the LAST 32 of it (the groups beyond the last full round, which run as narrower column tiles).  Every arithmetic mode of the build:
    f64 storage with the literal rule (the reference's arithmetic, 1e-9), f32 with the production rule and with the literal
    rule (5e-5); the fused launch and the fused dual-control launch (the control's correlations against the oracle too, and the
    real ones of the dual launch equal to the single launch: bit for bit in f32 - same chunk length, same order of summation -, to
    1e-12 in f64 where the dual kernel's chunks are shorter)."""
    w, ops = world, world["ops"]
    dev = w["dev"]
    import bench
    pcs = w["pcs"]
    emb = pcs[:, :2].contiguous()
    wide, _ = bench.sample_neighbors_device(emb, C // 5, 0.3, dev)
    assert wide.shape == (C, 3000)
    order = ops.hilbert_order(emb)
    NCHK = 128
    cells = torch.cat([order[p:p + 16] for p in (4000, 12000, 20000, 28000, 36000, 44000)] + [order[C - 32:]]).long()
    assert cells.numel() == NCHK
    assert C - 32 >= (C // 8 // 256) * 256 * 8 and C - 32 >= (-(-C // 6) // 256) * 256 * 6       # the last 32 sit in the tiled tail part (8- and 6-cell groups)
    # the graph and the pooled matrices from the count layers, in both storage types
    idx, dist = ops.knn_search(pcs, K)
    cS, cU, fS, fU, _ = bench_counts()
    conn = (dist > 0).double()
    wrow = torch.cat([torch.ones((C, 1), device=dev, dtype=torch.float64), conn], 1)
    wrow = (wrow / wrow.sum(1, keepdim=True)).contiguous()
    indices = torch.cat([torch.arange(C, device=dev, dtype=torch.int32)[:, None], idx], 1).contiguous()
    indptr = torch.arange(0, (C + 1) * (K + 1), K + 1, device=dev, dtype=torch.int64)
    indices, wrow = ops.canonical_graph_rows(indices, wrow)
    gen = torch.Generator(device=dev).manual_seed(5)
    d2_f32 = torch.randn((C, ops.padded_ld(G)), generator=gen, device=dev, dtype=torch.float32)       # a control with the shape of dmat_rndm; the
    d2_f32[:, G:] = 0                                                                                  # same values in both storage types
    got, ref_inputs = {}, None
    for dtype, name in ((torch.float64, "f64"), (torch.float32, "f32")):
        Sx, Ux = ops.knn_pool_counts(cS, cU, fS, fU, indptr, indices, wrow.to(dtype), dtype=dtype, validate=False)
        gam = ops.fit_slope(Ux, Sx)
        gam[~torch.isfinite(gam)] = 0.0
        d2 = ops.CellMatrix(d2_f32.to(dtype), G)
        if name == "f64":
            ref_inputs = (Sx, Ux, gam.double().cpu().numpy(), d2)
        rule_sets = (("literal", ops.RULES_PARTIAL),) if name == "f64" else (("production", ops.partial_rules_for(Sx, ops.SQRT, 1e-10)), ("literal", ops.RULES_PARTIAL))
        if name == "f32":
            assert rule_sets[0][1] == ops.RULES_PARTIAL_NOPSC
        for rname, rules in rule_sets:
            single = ops.coldeltacor_partial_fused(Sx, Ux, gam, None, wide, ops.SQRT, rules, 1e-10, order=order, validate=False)
            real, ctrl = ops.coldeltacor_partial_fused_dual(Sx, Ux, gam, None, d2, wide, ops.SQRT, rules, 1e-10, order=order, validate=False)
            fin = torch.isfinite(single)
            assert float(fin.float().mean()) > 0.999 and float(single[fin].abs().max()) <= 1 + 1e-5
            if name == "f32":                  # same chunk length as the single kernel -> same order of summation
                assert torch.equal(torch.nan_to_num(real, nan=7.0), torch.nan_to_num(single, nan=7.0)), "dual launch must reproduce the single launch bit for bit"
            else:                              # f64: the dual kernel walks 768-gene chunks, the single one 1024: summation order differs
                assert torch.equal(torch.isfinite(real), fin) and float((real[fin] - single[fin]).abs().max()) < 1e-12
            got[(name, rname)] = (single[cells].double().cpu().numpy(), ctrl[cells].double().cpu().numpy())
            del single, real, ctrl
        if name == "f32":
            del Sx, Ux, d2
    # ---- the oracle on the rows these cells touch (fp64 pooled matrices downloaded; the batch's cells are columns 0..15)
    Sx, Ux, g64, d2 = ref_inputs
    tol = {"f64": 1e-9, "f32": 5e-5}
    worst = {}
    for b in range(0, NCHK, 16):
        cs = cells[b:b + 16]
        nb = wide[cs].long().cpu().numpy()
        cs_np = cs.cpu().numpy()
        others = np.setdiff1d(np.unique(nb.ravel()), cs_np)
        rows = np.concatenate([cs_np, others])
        assert len(rows) < 20000                                # spatially adjacent cells share their 10 000 nearest: the sub-problem stays small
        col = np.full(C, -1, dtype=np.int64)
        col[rows] = np.arange(len(rows))
        e_sub = Sx.t[torch.as_tensor(rows, device=dev), :G].cpu().numpy().T.copy()          # (G, rows), the oracle's layout
        s, u = e_sub[:, :16], Ux.t[cs, :G].cpu().numpy().T
        Dv = (s + (u - g64[:, None] * s)) - s
        ixs = np.zeros((len(rows), nb.shape[1]), dtype=np.int64)
        ixs[:16] = col[nb]
        refs = []
        for dvals in (np.sign(Dv) * np.sqrt(np.abs(Dv) + 1e-10), d2.t[cs, :G].cpu().numpy().T):
            d_sub = np.zeros_like(e_sub)
            d_sub[:, :16] = dvals
            refs.append(oracle.coldeltacor_partial_compact(e_sub, d_sub, ixs, "sqrt", 1e-10, c0=0, c1=16)[:16])
            del d_sub
        for key, (real, ctrl) in got.items():
            for which, g_, ref in (("real", real[b:b + 16], refs[0]), ("control", ctrl[b:b + 16], refs[1])):
                ok = np.isfinite(ref)
                assert np.array_equal(np.isnan(g_), ~ok), (key, which)
                worst[key + (which,)] = max(worst.get(key + (which,), 0.0), float(np.abs(g_[ok] - ref[ok]).max()))
        del e_sub
    for key, v in worst.items():
        assert v <= tol[key[0]], worst


def test_fullsize_headline_arithmetic_against_the_oracle(world, oracle):
    """The bench's HEADLINE path at its full size - count layers -> vcy_knn_pool_counts in f64 -> fit_slope -> the fused f64 stage-D launch
    with the literal rule, scheduled along the Hilbert curve (bench.Pipeline with --dtype f64) - against the fp64 oracle on 128 whole
    cells x 250 neighbours x 30 000 genes: every pooled value of those cells and of the rows they touch to 1e-12, their gammas' inputs
    by construction, all 32 000 correlations to 1e-9 (measured 2e-13), NaN pattern equal; 64 of the cells come from the groups beyond
    the last full round of the launch (6-cell groups: the tiled tail)."""
    w, ops = world, world["ops"]
    dev = w["dev"]
    pcs, neigh = w["pcs"], w["neigh"]
    rng = np.random.default_rng(41)
    idx, dist = ops.knn_search(pcs, K)
    cS, cU, fS, fU, _ = bench_counts()
    wrow = torch.cat([torch.ones((C, 1), device=dev, dtype=torch.float64), (dist > 0).double()], 1)
    wrow = (wrow / wrow.sum(1, keepdim=True)).contiguous()
    indices = torch.cat([torch.arange(C, device=dev, dtype=torch.int32)[:, None], idx], 1).contiguous()
    indptr = torch.arange(0, (C + 1) * (K + 1), K + 1, device=dev, dtype=torch.int64)
    indices, wrow = ops.canonical_graph_rows(indices, wrow)
    Sx, Ux = ops.knn_pool_counts(cS, cU, fS, fU, indptr, indices, wrow, dtype=torch.float64, validate=False)
    gam = ops.fit_slope(Ux, Sx)
    gam[~torch.isfinite(gam)] = 0.0
    g64 = gam.double().cpu().numpy()
    order = ops.hilbert_order(pcs[:, :2].contiguous())
    assert ops.partial_rules_for(Sx, ops.SQRT, 1e-10) == ops.RULES_PARTIAL          # f64: the literal rule, always
    corr = ops.coldeltacor_partial_fused(Sx, Ux, gam, None, neigh, ops.SQRT, ops.RULES_PARTIAL, 1e-10, order=order, validate=False)
    tail0 = (-(-C // 6) // 256) * 256 * 6                   # first schedule position of the tiled tail (6-cell groups, 256 CUs)
    assert 0 < C - tail0 < 1536
    pos = np.concatenate([rng.choice(tail0, 64, replace=False), rng.choice(np.arange(tail0, C), 64, replace=False)])
    cells = order.cpu().numpy()[pos].astype(np.int64)
    worst_corr, worst_pool = 0.0, 0.0
    for b in range(0, len(cells), 64):
        cs = cells[b:b + 64]
        nb = neigh[torch.as_tensor(cs, device=dev)].long().cpu().numpy()
        others = np.setdiff1d(np.unique(nb.ravel()), cs)
        rows = np.concatenate([cs, others])
        col = np.full(C, -1, dtype=np.int64)
        col[rows] = np.arange(len(rows))
        rows_t = torch.as_tensor(rows, device=dev)
        e_sub = Sx.t[rows_t, :G].cpu().numpy().T.copy()
        # the pooled rows themselves, recomputed in fp64 from the count layers in the reference's order (ascending cell number)
        probe = rows[:: max(1, len(rows) // 24)][:24]
        for r in probe:
            js, ws = indices[r].long(), wrow[r].cpu().numpy()
            raw = cS.t[js, :G]
            raw = raw.double() if raw.dtype == torch.uint8 else (raw.to(torch.int32) & 0xFFFF).double()
            src = raw.cpu().numpy() * fS[js].cpu().numpy()[:, None]
            ref_row = np.zeros(G)
            for t in range(len(ws)):                        # the kernel's own order of summation: ascending cell number
                ref_row += ws[t] * src[t]
            got_row = Sx.t[r, :G].cpu().numpy()
            worst_pool = max(worst_pool, float(np.abs(got_row - ref_row).max() / max(1e-300, np.abs(ref_row).max())))
            # ... and through the ORACLE's own entry point (convolve_by_sparse_weights, neighbors.py:416-423) on the compact problem of this
            # row's 31 cells: row 0 of the weight matrix is the cell's weights, the other rows are unit rows (the oracle insists on rows summing to 1)
            from scipy import sparse
            nn = len(ws)
            Wsub = sparse.lil_matrix((nn, nn))
            Wsub.setdiag(1.0)
            Wsub[0, :] = ws
            ref_o = oracle.convolve_by_sparse_weights(np.ascontiguousarray(src.T), Wsub.tocsr())[:, 0]
            worst_pool = max(worst_pool, float(np.abs(got_row - ref_o).max() / max(1e-300, np.abs(ref_o).max())))
        s, u = e_sub[:, :len(cs)], Ux.t[torch.as_tensor(cs, device=dev), :G].cpu().numpy().T
        Dv = (s + (u - g64[:, None] * s)) - s
        d_sub = np.zeros_like(e_sub)
        d_sub[:, :len(cs)] = np.sign(Dv) * np.sqrt(np.abs(Dv) + 1e-10)
        ixs = np.zeros((len(rows), nb.shape[1]), dtype=np.int64)
        ixs[:len(cs)] = col[nb]
        ref = oracle.coldeltacor_partial_compact(e_sub, d_sub, ixs, "sqrt", 1e-10, c0=0, c1=len(cs))[:len(cs)]
        got = corr[torch.as_tensor(cs, device=dev)].cpu().numpy()
        ok = np.isfinite(ref)
        assert np.array_equal(np.isnan(got), ~ok)
        worst_corr = max(worst_corr, float(np.abs(got[ok] - ref[ok]).max()))
        del e_sub, d_sub
    assert worst_pool <= 1e-12 and worst_corr <= 1e-9, (worst_pool, worst_corr)
