"""Atlas-scale form of the path (BASELINE.json configs[4]): CSR count layers, pooling from sparse rows, block-streamed
stage D with sharded e.  Every kernel call goes through the C ABI on cuda:0.

Bars: the CSR pooling is BIT-IDENTICAL to the dense pooling of the densified layer (same arithmetic in the same order);
the streamed path in one block is bit-identical to the resident dense path, in several blocks equal up to the fp64
summation order of the fit moments (correlations 2e-6); spot checks against the CPU oracle at the f32 tolerances of
test_gpu_ops.py.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ops():
    import velocyto_amd
    from velocyto_amd import ops as _ops
    _ops.require_gpu()
    return _ops


def _random_counts(rng, C, G, density, big=False):
    a = rng.poisson(1.5, (C, G)) * (rng.random((C, G)) < density)
    if big:
        a[rng.integers(0, C, 5), rng.integers(0, G, 5)] = rng.integers(300, 60000, 5)      # forces uint16 storage
    a[min(3, C - 1)] = 0                                                                    # an empty row
    return a.astype(np.int64)


def test_csr_counts_container(ops):
    import scipy.sparse as sp
    rng = np.random.default_rng(0)
    for big in (False, True):
        a = _random_counts(rng, 70, 4500, 0.08, big)
        dense = ops.CountMatrix.from_genes_major(a.T.copy())
        csr = ops.CsrCounts.from_dense(dense)
        assert csr.C == 70 and csr.G == 4500 and csr.nnz == int((a != 0).sum())
        assert (csr.data.dtype == torch.int16) == big
        ref = sp.csr_matrix(a)
        assert np.array_equal(csr.indptr.cpu().numpy(), ref.indptr) and np.array_equal(csr.indices.cpu().numpy(), ref.indices)
        assert np.array_equal(csr.to_dense().as_int32().cpu().numpy(), a)
        assert np.array_equal(csr.row_sums().cpu().numpy(), a.sum(1))
        c2 = ops.CsrCounts.from_scipy(ref)
        assert torch.equal(c2.indptr, csr.indptr) and torch.equal(c2.indices, csr.indices) and torch.equal(c2.data, csr.data)
        sel = torch.tensor([5, 3, 69, 5, 0])
        sub = csr.rows(sel)
        assert np.array_equal(sub.to_dense().as_int32().cpu().numpy(), a[sel.numpy()])
        # slab table: lower bounds of every 2048-gene boundary inside each row
        sp_t = csr.slabptr.cpu().numpy()
        for r in (0, 3, 17):
            row = ref.indices[ref.indptr[r]:ref.indptr[r + 1]]
            assert np.array_equal(sp_t[r], np.searchsorted(row, np.arange(sp_t.shape[1]) * 2048))


@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("big", [False, True])
@pytest.mark.parametrize("G", [700, 2048, 5001])
def test_knn_pool_csr_bit_identical_to_dense(ops, dtype, big, G):
    rng = np.random.default_rng(G + big)
    C, k = 150, 9
    a = _random_counts(rng, C, G, 0.1, big)
    dense = ops.CountMatrix.from_genes_major(a.T.copy())
    csr = ops.CsrCounts.from_dense(dense)
    scale = torch.as_tensor(rng.gamma(4.0, 0.25, C))
    # ragged graph: row lengths 0 .. k + 70 (one row longer than a wave), self first where present
    lens = rng.integers(1, k + 2, C)
    lens[7], lens[11] = 0, k + 70
    indptr = np.concatenate([[0], np.cumsum(lens)])
    indices = np.concatenate([rng.choice(C, n, replace=False) for n in lens]).astype(np.int32)
    w = rng.random(indices.size)
    for maximum in (False, True):
        ref = ops.knn_pool_counts(dense, None, scale, None, indptr, indices, w, dtype=dtype, maximum=maximum)
        got = ops.knn_pool_csr(csr, scale, indptr, indices, w, dtype=dtype, maximum=maximum)
        assert torch.equal(got.t, ref.t), f"maximum={maximum}: CSR pooling differs from dense pooling"
    # a block of output cells with a schedule, written into rows of a larger buffer
    c0, n = 40, 64
    ip = indptr[c0:c0 + n + 1] - indptr[c0]
    ix, ww = indices[indptr[c0]:indptr[c0 + n]], w[indptr[c0]:indptr[c0 + n]]
    buf = ops.CellMatrix.empty(n + 5, G, dtype)
    buf.t.fill_(-1.0)
    order = torch.as_tensor(rng.permutation(n).astype(np.int32))
    ops.knn_pool_csr(csr, scale, ip, ix, ww, dtype=dtype, cell0=c0, C_out=n, out=buf, order=order, maximum=True)
    assert torch.equal(buf.t[:n], ops.knn_pool_counts(dense, None, scale, None, indptr, indices, w, dtype=dtype, maximum=True).t[c0:c0 + n])
    assert bool((buf.t[n:] == -1.0).all())
    # and the product itself in fp64 (neighbors.py:416-423 on S_sz = scale * counts; the oracle's own entry point insists
    # on rows of w summing to one, which this ragged graph - empty row included - does not)
    import scipy.sparse as sp
    W = sp.csr_matrix((w, indices, indptr), shape=(C, C))
    ref_o = np.asarray(W.dot((a * scale.numpy()[:, None]).astype(np.float64))).T
    got = ops.knn_pool_csr(csr, scale, indptr, indices, w, dtype=dtype).to_genes_major()
    np.testing.assert_allclose(got, ref_o, rtol=2e-6 if dtype == "float32" else 1e-12, atol=1e-6 if dtype == "float32" else 1e-12)


@pytest.mark.parametrize("nnz", [0, 1, 3, 4, 5])
def test_knn_pool_csr_layers_with_a_handful_of_nonzeros(ops, nnz):
    """A CSR layer holding fewer non-zeros than one 16-byte quad of the pooling kernel (its loads are pulled back to end inside
    the arrays): the container pads the stored arrays to 4 elements (contract of vcy_knn_pool_csr, velocyto_hip.h), the padding
    belongs to no row, and pooling equals the dense kernel bit for bit - also when the few non-zeros sit in the LAST row."""
    rng = np.random.default_rng(nnz)
    C, G, k = 40, 700, 5
    a = np.zeros((C, G), dtype=np.int64)
    spots = [(C - 1, G - 1), (C - 1, 3), (0, 0), (17, 350), (C - 1, 100)][:nnz]
    for r, g in spots:
        a[r, g] = rng.integers(1, 200)
    dense = ops.CountMatrix.from_genes_major(a.T.copy())
    csr = ops.CsrCounts.from_dense(dense)
    assert csr.nnz == nnz and csr.indices.numel() == nnz and csr.data.numel() == nnz and csr._istore.numel() >= 4
    assert np.array_equal(csr.to_dense().as_int32().cpu().numpy(), a) and np.array_equal(csr.row_sums().cpu().numpy(), a.sum(1))
    indptr = np.arange(0, C * k + 1, k)
    indices = np.concatenate([rng.choice(C, k, replace=False) for _ in range(C)]).astype(np.int32)
    indices[(C - 1) * k] = C - 1                              # somebody pools the last row
    w = rng.random(indices.size)
    scale = torch.as_tensor(rng.gamma(4.0, 0.25, C))
    for dtype in ("float32", "float64"):
        for maximum in (False, True):
            ref = ops.knn_pool_counts(dense, None, scale, None, indptr, indices, w, dtype=dtype, maximum=maximum)
            got = ops.knn_pool_csr(csr, scale, indptr, indices, w, dtype=dtype, maximum=maximum)
            assert torch.equal(got.t, ref.t)
    assert np.array_equal(csr.rows(torch.tensor([C - 1, 0, 17])).to_dense().as_int32().cpu().numpy(), a[[C - 1, 0, 17]])


def _dense_reference(ops, atlas, cS, cU, fS, fU, pcs, emb, k, n_neighbors, frac):
    """The resident dense path on the same data: knn_pool_counts -> fit_slope -> fused stage D with a full-height e."""
    C = cS.C
    idx, dist = ops.knn_search(pcs, k, include_self=False)
    conn = (dist > 0).float()
    w = torch.cat([torch.ones((C, 1), device=idx.device), conn], 1)
    w = w / w.sum(1, keepdim=True)
    ind = torch.cat([torch.arange(C, device=idx.device, dtype=torch.int32)[:, None], idx], 1)
    ind, w = ops.canonical_graph_rows(ind, w)                # rows by cell number: the order every device-built graph pools in
    ptr = torch.arange(0, (C + 1) * (k + 1), k + 1, device=idx.device, dtype=torch.int64)
    Sx, Ux = ops.knn_pool_counts(cS.to_dense(), cU.to_dense(), fS, fU, ptr, ind, w, dtype=torch.float32, validate=False)
    gamma = ops.fit_slope_from_moments(ops.fit_slope_moments(Ux, Sx))
    gamma[~torch.isfinite(gamma)] = 0.0                      # analysis.py:1260
    neigh = atlas.sample_neighbors(emb.double(), 0, C, n_neighbors, frac)
    corr = ops.coldeltacor_partial_fused(Sx, Ux, gamma, None, neigh, ops.SQRT, ops.partial_rules_for(Sx, ops.SQRT, 1e-10), 1e-10, validate=False)
    return Sx, Ux, gamma, neigh, corr


def test_atlas_path_equals_resident_dense_path(ops):
    from velocyto_amd import atlas
    dev = ops.require_gpu()
    C, G, k = 6000, 3000, 12
    cS, cU, totS, totU, pcs, emb = atlas.synth_atlas(C, G, 12, dev, density=0.08)
    assert 0.06 < cS.nnz / (C * G) < 0.10
    fS, fU = atlas.size_factors(totS, totU, C)
    Sx, Ux, gamma, neigh, corr = _dense_reference(ops, atlas, cS, cU, fS, fU, pcs, emb, k, 100, 0.5)
    one = atlas.AtlasPath(cS, cU, fS, fU, pcs, emb, k=k, n_neighbors=100, sampled_fraction=0.5, block_cells=0)
    c1 = one.run().clone()
    assert torch.equal(one.neigh, neigh)
    e_buf, Ux_b = one._resident
    assert torch.equal(e_buf.t[:C], Sx.t) and torch.equal(Ux_b.t, Ux.t), "pooled matrices from CSR differ from the dense path"
    assert torch.equal(one.gamma, gamma)
    assert torch.equal(torch.nan_to_num(c1, nan=7.0), torch.nan_to_num(corr, nan=7.0)), "one-block streamed path must be bit-identical"
    many = atlas.AtlasPath(cS, cU, fS, fU, pcs, emb, k=k, n_neighbors=100, sampled_fraction=0.5, block_cells=1400)
    assert len(many.blocks()) == 5
    c5 = many.run()
    torch.testing.assert_close(many.gamma, gamma, rtol=2e-6, atol=1e-9)
    fin = torch.isfinite(corr)
    assert torch.equal(torch.isfinite(c5), fin)
    assert float((c5[fin] - corr[fin]).abs().max()) <= 2e-6
    assert many.peak_block_bytes < one.peak_block_bytes / 2        # the point of streaming: O(block) dense memory


def test_atlas_path_in_f64_with_garbage_in_the_staging_buffers(ops):
    """The f64 atlas pass (the bench's default arithmetic) over several blocks, with the staging buffers holding what a caching allocator
    may hand out - bit patterns that read as 1e300 / inf / NaN in f64 - before the first block is pooled: the square-root domain check looks
    at the pooled blocks (the matrix itself), never at rows of the staging buffer that are not written yet, and the blocked run equals
    the one-block run."""
    from velocyto_amd import atlas
    dev = ops.require_gpu()
    C, G, k = 4000, 2100, 10
    cS, cU, totS, totU, pcs, emb = atlas.synth_atlas(C, G, 10, dev, density=0.08)
    fS, fU = atlas.size_factors(totS, totU, C)
    one = atlas.AtlasPath(cS, cU, fS, fU, pcs, emb, k=k, n_neighbors=80, sampled_fraction=0.5, block_cells=0, dtype=torch.float64)
    c1 = one.run().clone()
    many = atlas.AtlasPath(cS, cU, fS, fU, pcs, emb, k=k, n_neighbors=80, sampled_fraction=0.5, block_cells=900, dtype=torch.float64)
    assert len(many.blocks()) >= 4
    many._plan_blocks()                                                  # (the staging buffers are allocated with the block plan)
    many._ebuf.t[:, :G] = 1e308                                          # (the zero padding beyond G is the buffers' invariant: left alone)
    many._ebuf.t[::7, :G] = float("nan")
    many._ubuf.t[:, :G] = float("inf")
    c5 = many.run()
    fin = torch.isfinite(c1)
    assert torch.equal(torch.isfinite(c5), fin)
    assert float((c5[fin] - c1[fin]).abs().max()) <= 1e-12
    torch.testing.assert_close(many.gamma, one.gamma, rtol=1e-6, atol=1e-12)
    # a matrix that really leaves the domain is refused, by name
    big = atlas.AtlasPath(cS, cU, fS * 1e40, fU, pcs, emb, k=k, n_neighbors=80, sampled_fraction=0.5, block_cells=900, dtype=torch.float64)
    with pytest.raises(ValueError, match="outside the supported range"):
        big.run()
    torch.cuda.synchronize()


ARGS = ["--workload", "cfg5", "--no-cpu-baseline", "--cells", "9000", "--genes", "2100", "--n-neighbors", "100", "--k", "12", "--pca-dims", "10",
        "--steps", "1", "--warmup", "0"]


def _run_bench(world, dump, extra=(), port=29801, backend="gloo"):
    env = dict(os.environ, VCY_SINGLE_DEVICE="1", VCY_DIST_BACKEND=backend, VCY_FORCE_COLLECTIVES="1", MASTER_PORT=str(port), MASTER_ADDR="127.0.0.1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    for key in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(key, None)
    if world == 1:
        cmd = [sys.executable, "bench.py", "--gpus", "1", *ARGS, "--dump", dump, *extra]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), "bench.py", "--gpus", str(world), *ARGS, "--dump", dump, *extra]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    last = [l for l in r.stdout.strip().splitlines() if l.strip()][-1]
    assert last.startswith("{") and '"n_gpus": %d' % world in last and '"e_sharded": true' in last, last[:300]
    return dict(np.load(dump))


@pytest.mark.parametrize("world,extra", [(2, ()), (3, ()), (2, ("--block-cells", "1000"))])
def test_atlas_sharded_ranks_equal_one_rank(tmp_path, world, extra):
    """2 and 3 ranks on one GPU (gloo transport): count-row halo exchange, pooling of the halo e rows, sharded stage D and the
    all-gather of the correlation rows reproduce the one-rank run."""
    from velocyto_amd import ops
    ops.require_gpu()
    one = _run_bench(1, str(tmp_path / "one.npz"), port=29801 + world)
    many = _run_bench(world, str(tmp_path / "many.npz"), extra, port=29821 + world + len(extra))
    assert np.array_equal(one["neigh"], many["neigh"])
    np.testing.assert_allclose(many["gamma"], one["gamma"], rtol=2e-6, atol=1e-9)
    fin = np.isfinite(one["corr"])
    assert np.array_equal(np.isfinite(many["corr"]), fin)
    np.testing.assert_allclose(many["corr"][fin], one["corr"][fin], atol=2e-6)


@pytest.mark.parametrize("C,block_cells,nblocks,dtype", [(200_000, 50_000, 4, "float32"), (1_000_000, "auto", 3, "float32"), (1_000_000, "auto", 3, "float64")])
def test_atlas_fullsize_cells_30k_genes(ops, oracle, C, block_cells, nblocks, dtype):
    """BASELINE.json configs[4]: 30 000 genes, CSR layers at ~8 % density, streamed over cell blocks - at 200 000 cells in four
    blocks of 50 000 (dense Sx/Ux of the whole dataset would be 48 GB; a block holds ~15 GB) and at the STATED size, 1 000 000
    cells, on one MI355X in the default blocks (three of 366 350 cells on 288 GB; the dense f32 layers alone would be 240 GB).
      * pooling from CSR == pooling from the densified layers, bit for bit, on cell blocks from both ends of the dataset;
      * the exact kNN graph (projection-pruned from 100 000 cells on) == brute force on sampled queries;
      * gamma against fp64 column sums of the pooled blocks; correlations of sampled cells against the fp64 oracle on the
        rows they touch (f32 storage: 5e-5; f64 - the headline arithmetic, at the stated size - 1e-9); size-independent
        properties of the whole result."""
    from velocyto_amd import atlas
    dev = ops.require_gpu()
    tdt = getattr(torch, dtype)
    f64 = dtype == "float64"
    if C > 500_000 and torch.cuda.mem_get_info()[1] < 250e9:
        pytest.skip("the 1M-cell pass with its default blocks needs the memory of one MI355X")
    G, k = 30_000, 30
    cS, cU, totS, totU, pcs, emb = atlas.synth_atlas(C, G, 30, dev, density=0.08)
    dens = cS.nnz / (C * G)
    assert 0.07 < dens < 0.09, dens
    fS, fU = atlas.size_factors(totS, totU, C)
    if block_cells == "auto":                                         # what `bench.py --workload cfg5` picks from the free HBM
        block_cells = atlas.auto_block_cells(C, C, G, dev, 8 if f64 else 4)
    path = atlas.AtlasPath(cS, cU, fS, fU, pcs, emb, k=k, n_neighbors=500, sampled_fraction=0.5, block_cells=block_cells, dtype=tdt)
    assert path.nrndm == 250 and (len(path.blocks()) == nblocks if C <= 500_000 else 2 <= len(path.blocks()) <= (12 if f64 else 6)), path.blocks()
    corr = path.run()
    assert corr.shape == (C, 250)
    # ---- A: the graph the pass pooled with (pruned search from 100 000 cells on) against brute force on 2048 spread queries
    q = (torch.arange(2048, device=dev) * (C / 2048)).long()
    bi, bd = ops.knn_query(pcs, pcs[q], k + 1)                         # the query itself comes back in column 0 (distance 0)
    assert bool((bd[:, 0] == 0).all())
    assert torch.equal(torch.sort(bi.long(), 1).values, torch.sort(path.g_idx[q].long(), 1).values), "pruned kNN graph differs from brute force"
    # ---- A: CSR pooling against the dense kernel on the densified count rows the block touches, three blocks of 4096 cells
    for b0 in (0, C // 2 + 1000, C - 4096):
        rows = slice(b0, b0 + 4096)
        gi, gw = path.g_idx[rows], path.g_w[rows]
        sel = torch.unique(gi.reshape(-1).long())                       # count rows the block gathers (ascending); densify only those
        loc = torch.searchsorted(sel, gi.reshape(-1).long()).to(torch.int32)
        dS, dU = path.cS.rows(sel).to_dense(), path.cU.rows(sel).to_dense()
        ptr = torch.arange(0, 4097 * (k + 1), k + 1, device=dev, dtype=torch.int64)
        ref_S, ref_U = ops.knn_pool_counts(dS, dU, path.fS[sel], path.fU[sel], ptr, loc, gw.reshape(-1), dtype=tdt, C_out=4096, validate=False)
        got_S = ops.CellMatrix.empty(4096, G, tdt)
        got_U = ops.CellMatrix.empty(4096, G, tdt)
        path._pool(path.cS, path.fS, rows, got_S)
        path._pool(path.cU, path.fU, rows, got_U)
        assert torch.equal(got_S.t, ref_S.t) and torch.equal(got_U.t, ref_U.t), f"block at {b0}: CSR pooling differs from dense pooling"
        del dS, dU, ref_S, ref_U, got_S, got_U
    # ---- B: gamma = max(0, <Sx,Ux>/<Sx,Sx>) over ALL cells, re-derived in fp64 from freshly pooled blocks
    sxx = torch.zeros(G, dtype=torch.float64, device=dev)
    sxy = torch.zeros(G, dtype=torch.float64, device=dev)
    for b0 in range(0, C, 20_000):
        n = min(20_000, C - b0)
        bS, bU = ops.CellMatrix.empty(n, G, tdt), ops.CellMatrix.empty(n, G, tdt)
        path._pool(path.cS, path.fS, slice(b0, b0 + n), bS)
        path._pool(path.cU, path.fU, slice(b0, b0 + n), bU)
        sxx += (bS.t[:, :G].double() ** 2).sum(0)
        sxy += (bS.t[:, :G].double() * bU.t[:, :G].double()).sum(0)
    ref_g = torch.clamp(sxy / sxx, min=0)
    okg = sxx > 0
    torch.testing.assert_close(path.gamma.double()[okg], ref_g[okg], rtol=2e-6, atol=1e-12)      # (gamma is a float32 output)
    # ---- D: properties of the whole result
    fin = torch.isfinite(corr)
    assert fin.float().mean().item() > 0.999 and corr[fin].abs().max().item() <= 1 + 1e-5
    assert bool((path.neigh[:, 1:] > path.neigh[:, :-1]).all()) and int(path.neigh.max()) < C
    # ---- D: sampled cells against the fp64 oracle (first / last block, block boundaries)
    g64 = path.gamma.double().cpu().numpy()
    for c in (17, 49_999, 50_000, 123_456, C // 2 + 1, path.blocks()[1][0] - 1, path.blocks()[1][0], C - 3):     # incl. both sides of a block boundary
        nb = path.neigh[c].long()
        rows = torch.cat([torch.tensor([c], device=dev), nb])
        eS = ops.CellMatrix.empty(rows.numel(), G, tdt)
        path._pool(path.cS, path.fS, rows, eS)
        uC = ops.CellMatrix.empty(1, G, tdt)
        path._pool(path.cU, path.fU, rows[:1], uC)
        e_sub = eS.t[:, :G].double().cpu().numpy().T
        s, u = e_sub[:, 0], uC.t[0, :G].double().cpu().numpy()
        Dv = (s + (u - g64 * s)) - s
        d_sub = np.zeros_like(e_sub)
        d_sub[:, 0] = np.sign(Dv) * np.sqrt(np.abs(Dv) + 1e-10)
        ixs = np.zeros((e_sub.shape[1], nb.numel()), dtype=np.int64)
        ixs[0] = np.arange(1, nb.numel() + 1)
        ref = oracle.coldeltacor_partial_compact(e_sub, d_sub, ixs, "sqrt", 1e-10, c0=0, c1=1)[0]
        got = corr[c].cpu().numpy()
        okc = np.isfinite(ref)
        np.testing.assert_allclose(got[okc], ref[okc], atol=1e-9 if f64 else 5e-5)


def test_atlas_path_on_the_rccl_transport(tmp_path):
    """The atlas path's collectives (all-gather of pcs / embedding, halo masks, graph-row and ragged count-row all-to-alls,
    all-reduce of the fit moments) on backend "nccl" (= RCCL) at world size 1, against the gloo run of the same problem."""
    from velocyto_amd import ops
    ops.require_gpu()
    ref = _run_bench(1, str(tmp_path / "gloo.npz"), port=29861)
    got = _run_bench(1, str(tmp_path / "rccl.npz"), port=29862, backend="nccl")
    assert np.array_equal(ref["neigh"], got["neigh"]) and np.array_equal(ref["gamma"], got["gamma"])
    fin = np.isfinite(ref["corr"])
    assert np.array_equal(np.isfinite(got["corr"]), fin) and np.array_equal(got["corr"][fin], ref["corr"][fin])


def test_loom_file_to_atlas_path(ops, tmp_path):
    """A .loom file straight into the atlas path: layers read in hyperslabs of cells into device CSR (never dense), size
    factors from the CSR row sums, AtlasPath in three blocks - against the dense facade-level route on the same file."""
    import velocyto_amd
    from velocyto_amd import atlas, loom_io
    dev = ops.require_gpu()
    rng = np.random.default_rng(21)
    G, C, k = 900, 700, 10
    lam = rng.gamma(0.3, 1.0, (G, 1)) * rng.gamma(2.0, 0.5, (1, C))
    S, U = rng.poisson(lam).astype(np.uint16), rng.poisson(0.4 * lam).astype(np.uint16)
    S[3, 5] = 3000                                                     # beyond a byte: uint16 counts on the device
    path = str(tmp_path / "atlas.loom")
    loom_io.write_loom(path, {"spliced": S, "unspliced": U, "ambiguous": np.zeros_like(S)}, {"CellID": np.arange(C)}, {"Gene": np.arange(G)})
    cS, cU = loom_io.read_layer_csr(path, "spliced", cell_block=256), loom_io.read_layer_csr(path, "unspliced", cell_block=256)
    assert cS.data.dtype == torch.int16 and cU.data.dtype == torch.uint8 and cS.nnz == int((S != 0).sum())
    fS, fU = atlas.size_factors(cS.row_sums(), cU.row_sums(), C)
    pcs = torch.as_tensor(rng.normal(size=(C, 6)), device=dev)
    emb = pcs[:, :2].contiguous()
    a = atlas.AtlasPath(cS, cU, fS, fU, pcs, emb, k=k, n_neighbors=60, sampled_fraction=0.5, block_cells=250, knn="brute")
    corr = a.run()
    # the dense route: facade normalisation (analysis.py:535-582) gives the same size factors; pooling + fit + fused stage D
    vlm = velocyto_amd.analysis.VelocytoLoom(path, dtype="float32")
    vlm.normalize("both", size=True, log=False)
    np.testing.assert_allclose((vlm.S_sz[:, 7] / np.maximum(S[:, 7], 1))[S[:, 7] > 0], float(fS[7]), rtol=1e-6)
    Sx, Ux, gamma, neigh, ref = _dense_reference(ops, atlas, cS, cU, fS, fU, pcs, emb, k, 60, 0.5)
    assert torch.equal(a.neigh, neigh)
    torch.testing.assert_close(a.gamma, gamma, rtol=2e-6, atol=1e-9)
    fin = torch.isfinite(ref)
    assert torch.equal(torch.isfinite(corr), fin) and float((corr[fin] - ref[fin]).abs().max()) <= 2e-6
