"""GPU parity tests, kernel level: every call goes through the C ABI (velocyto_amd.ops ->
libvelocyto_hip.so) on cuda:0 and is compared with the CPU oracle / the golden vectors
recorded from the reference.

Tolerances (stated per dtype):
  f64 storage: correlations atol 1e-10 (raw-moment single pass vs the reference's centred
               two-pass), pooled matrices rtol 1e-12, quantiles exact selection (rtol 1e-14).
  f32 storage: correlations atol 5e-5, pooled/velocity matrices rtol 2e-6 (+atol 1e-6),
               gammas rtol 1e-5.
Integer outputs (neighbour indices, balanced graph) are bit-exact.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ops():
    import velocyto_amd
    from velocyto_amd import ops as _ops
    _ops.require_gpu()
    return _ops


CORR_ATOL = {"float64": 1e-10, "float32": 5e-5}


def _nan_close(got, ref, atol, skip=None):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    bad = np.isnan(ref) if skip is None else (np.isnan(ref) | skip)
    must_be_nan = np.isnan(ref) if skip is None else (np.isnan(ref) & ~skip)
    assert np.isnan(got[must_be_nan]).all()
    np.testing.assert_allclose(got[~bad], ref[~bad], atol=atol, rtol=0)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_transpose_roundtrip(ops, dtype):
    rng = np.random.default_rng(1)
    for G, C in ((1, 1), (3, 130), (257, 64), (1000, 333)):
        a = rng.normal(size=(G, C))
        m = ops.CellMatrix.from_genes_major(a, dtype)
        assert m.C == C and m.G == G and m.ld % 64 == 0
        assert float(m.t[:, G:].abs().sum()) == 0.0          # zero padding
        back = m.to_genes_major()
        np.testing.assert_allclose(back, a, rtol=(1e-6 if dtype == "float32" else 0), atol=0)
        np.testing.assert_array_equal(m.to_genes_major(order="F"), back)
        # Fortran-ordered input (what the reference's pooling returns) takes the no-transpose path
        m2 = ops.CellMatrix.from_genes_major(np.asfortranarray(a), dtype)
        assert torch.equal(m2.t, m.t)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("key,transform,psc_key", [
    ("partial_linear", "linear", None), ("partial_sqrt_a", "sqrt", "psc_a"), ("partial_sqrt_b", "sqrt", "psc_b"),
    ("partial_log10_a", "log10", "psc_a"), ("partial_log10_b", "log10", "psc_b")])
def test_coldeltacor_partial_golden(ops, golden, dtype, key, transform, psc_key):
    g = golden("coldeltacor")
    psc = float(g[psc_key]) if psc_key else 0.0
    e, d = ops.CellMatrix.from_genes_major(g["e"], dtype), ops.CellMatrix.from_genes_major(g["d"], dtype)
    comp = ops.coldeltacor_partial(e, d, g["ixs"], ops.TRANSFORMS[transform], ops.RULES_PARTIAL, psc)
    dense = ops.scatter_rows(comp, g["ixs"], e.C).cpu().numpy()
    ref = g[key]
    degenerate = np.zeros(ref.shape, bool)
    degenerate[3, 7] = True                          # identical cells / a cell paired with itself are zero-variance
    degenerate |= np.eye(ref.shape[0], dtype=bool)   # columns: exact NaN for sqrt/linear, NaN-or-rounding-noise for log10
    if transform in ("sqrt", "linear"):
        assert np.isnan(dense[3, 7]) and np.isnan(dense[5, 5])
    # (f32, log10, psc = 1e-10: log10(0 + 1e-10) = -10 on every agreeing gene used to swamp the f32 raw moments; the
    #  kernels now accumulate A - f(0), csrc/coldeltacor.hip "Shifted moments", and the case is held to the f32 tolerance)
    np.testing.assert_allclose(dense[~degenerate], ref[~degenerate], atol=CORR_ATOL[dtype])
    assert (dense[(ref == 0) & ~degenerate] == 0).all()        # cells that were never listed stay exactly zero


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("key,transform,psc_key", [
    ("full_linear", "linear", None), ("full_sqrt_a", "sqrt", "psc_a"), ("full_sqrt_b", "sqrt", "psc_b"),
    ("full_log10_b", "log10", "psc_b"), ("full_log10_a", "log10", "psc_a")])
def test_coldeltacor_full_golden(ops, golden, dtype, key, transform, psc_key):
    g = golden("coldeltacor")
    psc = float(g[psc_key]) if psc_key else 0.0
    e, d = ops.CellMatrix.from_genes_major(g["e"], dtype), ops.CellMatrix.from_genes_major(g["d"], dtype)
    rm = ops.coldeltacor_full(e, d, ops.TRANSFORMS[transform], psc).cpu().numpy()
    ref = g[key]
    C = ref.shape[0]
    degenerate = np.eye(C, dtype=bool)
    degenerate[3, 7] = degenerate[7, 3] = True
    np.testing.assert_allclose(rm[~degenerate], ref[~degenerate], atol=CORR_ATOL[dtype])
    if transform in ("linear", "sqrt"):      # identical cells 3 and 7: zero variance of A -> NaN, as the reference's 0 * inf (also on the matrix-core route)
        assert np.isnan(rm[3, 7]) and np.isnan(rm[7, 3]) and np.isnan(rm[5, 5])
    # row-block + accumulate semantics (rm[c,i] += ...)
    blk = ops.coldeltacor_full(e, d, ops.TRANSFORMS[transform], psc, cell0=8, C_out=17)
    if transform == "linear":      # matrix-core route: a row block is tiled differently from the full product
        np.testing.assert_allclose(blk.cpu().numpy()[~degenerate[8:25]], rm[8:25][~degenerate[8:25]], atol=1e-12 if dtype == "float64" else 1e-6)
    else:
        np.testing.assert_array_equal(blk.cpu().numpy()[~degenerate[8:25]], rm[8:25][~degenerate[8:25]])
    acc = torch.ones((17, C), dtype=blk.dtype, device=blk.device)
    ops.coldeltacor_full(e, d, ops.TRANSFORMS[transform], psc, cell0=8, C_out=17, rm=acc, accumulate=True)
    np.testing.assert_allclose(acc.cpu().numpy()[~degenerate[8:25]], 1 + rm[8:25][~degenerate[8:25]], atol=1e-6)


@pytest.mark.parametrize("dtype,G", [("float64", 2500), ("float32", 2500), ("float32", 40000), ("float64", 21001)])
@pytest.mark.parametrize("transform,psc", [("sqrt", 1e-10), ("log10", 1.0), ("linear", 0.0)])
def test_coldeltacor_partial_vs_oracle(ops, oracle, dtype, G, transform, psc):
    """Seeded random case incl. gene counts that force several LDS chunks and ragged tails."""
    rng = np.random.default_rng(G)
    C, nr = 48, 20
    e = rng.gamma(2.0, 1.0, (G, C)) * (rng.random((G, C)) < 0.6)
    d = rng.normal(0, 1, (G, C))
    ixs = np.stack([rng.choice(C, nr, replace=False) for _ in range(C)])
    em, dm = ops.CellMatrix.from_genes_major(e, dtype), ops.CellMatrix.from_genes_major(d, dtype)
    got = ops.coldeltacor_partial(em, dm, ixs, ops.TRANSFORMS[transform], ops.RULES_PARTIAL, psc).cpu().numpy()
    ref = oracle.coldeltacor_partial_compact(e, d, ixs, transform, psc)
    self_pair = ixs == np.arange(C)[:, None]
    assert np.isnan(got[self_pair]).all() or transform == "log10"
    np.testing.assert_allclose(got[~self_pair], ref[~self_pair], atol=CORR_ATOL[dtype])
    assert np.all(np.abs(got[~self_pair]) <= 1 + 1e-5)
    # scheduling order must not change results; row blocks must agree with the full call
    order = torch.from_numpy(rng.permutation(C).astype(np.int32))
    got2 = ops.coldeltacor_partial(em, dm, ixs, ops.TRANSFORMS[transform], ops.RULES_PARTIAL, psc, order=order).cpu().numpy()
    np.testing.assert_array_equal(got2, got)
    # a 21-cell block runs the one-cell-per-workgroup kernel, the full call the grouped one: same sums, different order
    got3 = ops.coldeltacor_partial(em, dm, ixs[10:31], ops.TRANSFORMS[transform], ops.RULES_PARTIAL, psc, cell0=10).cpu().numpy()
    np.testing.assert_allclose(got3[~self_pair[10:31]], got[10:31][~self_pair[10:31]], atol=1e-12 if dtype == "float64" else 2e-5)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("transform", ["sqrt", "log10", "linear"])
def test_coldeltacor_partial_fused_equals_two_kernels(ops, dtype, transform):
    """Velocity chain folded into the correlation kernel == velocity_chain followed by coldeltacor_partial, bit for bit."""
    rng = np.random.default_rng(17)
    G, C, nr = 3100, 72, 24
    Sx = rng.gamma(2.0, 1.0, (G, C)) * (rng.random((G, C)) < 0.7)
    Ux = rng.gamma(1.0, 1.0, (G, C)) * (rng.random((G, C)) < 0.6)
    gam = torch.from_numpy(rng.random(G).astype(np.float32))
    q = torch.from_numpy((0.1 * rng.random(G)).astype(np.float32))
    ixs = np.stack([rng.choice(C, nr, replace=False) for _ in range(C)])
    S, U = ops.CellMatrix.from_genes_major(Sx, dtype), ops.CellMatrix.from_genes_major(Ux, dtype)
    tr = ops.TRANSFORMS[transform]
    psc = 1e-10 if transform == "sqrt" else (1.0 if transform == "log10" else 0.0)
    for qq, dts, udt in ((q, 1.0, 1.0), (None, 0.5, 2.0)):
        dm = ops.velocity_chain(S, U, gam, qq, want=("dmat",), dt_shift=dts, used_dt=udt, transform=tr, psc=psc)["dmat"]
        ref = ops.coldeltacor_partial(S, dm, ixs, tr, ops.RULES_PARTIAL, psc)
        got = ops.coldeltacor_partial_fused(S, U, gam, qq, ixs, tr, ops.RULES_PARTIAL, psc, dt_shift=dts, used_dt=udt)
        assert torch.equal(torch.nan_to_num(got, nan=7.0), torch.nan_to_num(ref, nan=7.0))
    part = ops.coldeltacor_partial_fused(S, ops.CellMatrix(U.t[16:64].contiguous(), G), gam, None, ixs[16:64], tr, ops.RULES_PARTIAL, psc,
                                         dt_shift=0.5, used_dt=2.0, cell0=16, u_row0=16)
    np.testing.assert_allclose(torch.nan_to_num(part, nan=7.0).cpu().numpy(), torch.nan_to_num(got[16:64], nan=7.0).cpu().numpy(),
                               atol=1e-12 if dtype == "float64" else 2e-5)
    # too few cells for the grouped kernel: the one-cell-per-workgroup kernel builds d[c] the same way (same values,
    # other summation order)
    S8, U8 = ops.CellMatrix(S.t[:8].contiguous(), G), ops.CellMatrix(U.t[:8].contiguous(), G)
    small = ops.coldeltacor_partial_fused(S8, U8, gam, q, ixs[:8] % 8, tr, ops.RULES_PARTIAL, psc)
    dm8 = ops.velocity_chain(S8, U8, gam, q, want=("dmat",), transform=tr, psc=psc)["dmat"]
    ref8 = ops.coldeltacor_partial(S8, dm8, ixs[:8] % 8, tr, ops.RULES_PARTIAL, psc)
    assert torch.equal(torch.nan_to_num(small, nan=7.0), torch.nan_to_num(ref8, nan=7.0))


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_coldeltacor_full_linear_gemm_route(ops, oracle, dtype):
    """The all-pairs linear variant as two contractions over the genes on the f64 matrix cores (vcy_coldeltacor_full_linear: Pearson
    epilogue fused, no library GEMM) == the element-wise kernel == the oracle (speedboosted.pyx:13-87), to the SAME tolerance on every
    pair: nearly identical cells (where the expanded sum of squares cancels) are re-evaluated by the entry's repair launch in the
    reference's centred difference form, exact duplicates and a constant d_c give the reference's NaN (0 * inf) and nothing else does;
    row blocks (cell0 / C_out), ragged tiles and `rm +=` as well."""
    rng = np.random.default_rng(31)
    G, C = 700, 130
    e, d = rng.gamma(2.0, 1.0, (G, C)), rng.normal(size=(G, C))
    near = {}
    for k, noise in enumerate((1e-3, 1e-4, 1e-5, 1e-6, 1e-7, 1e-8, 1e-9)):     # cell 21 + 2 k is a copy of cell 20 + 2 k up to `noise`
        e[:, 21 + 2 * k] = e[:, 20 + 2 * k] + noise * rng.normal(size=G)
        near[(20 + 2 * k, 21 + 2 * k)] = noise
    e[:, 5] = e[:, 4] * (1.0 + 1e-7)                        # a rescaled copy: A = 1e-7 e_c, its mean far from zero
    e[:, 9] = e[:, 8]                                        # exact duplicate
    d[:, 60] = 0.0                                           # a constant d_c: the whole row is 0 * inf in the reference
    d[:, 61] = 3.0 + 1e-5 * rng.normal(size=G)              # a nearly constant one (variance 1e-11 of its sum of squares): finite in the reference
    if dtype == "float32":                                   # the stored values are the inputs: the oracle sees what the kernel sees
        e, d = e.astype(np.float32).astype(np.float64), d.astype(np.float32).astype(np.float64)
    want = oracle.coldeltacor(e, d, "linear", 0.0)
    E, D = ops.CellMatrix.from_genes_major(e, dtype), ops.CellMatrix.from_genes_major(d, dtype)
    got = ops.coldeltacor_full(E, D, ops.LINEAR).cpu().numpy()
    ops.FULL_LINEAR_MFMA = False
    try:
        kern = ops.coldeltacor_full(E, D, ops.LINEAR).cpu().numpy()
    finally:
        ops.FULL_LINEAR_MFMA = True
    # the NaN pattern is the reference's: the diagonal, the duplicate pair, the row of the constant d_c - nothing else
    nan_want = np.isnan(want)
    expect = np.eye(C, dtype=bool)
    expect[8, 9] = expect[9, 8] = True
    expect[60, :] = True
    for (a, b) in near:                                      # (f32 storage may round the smallest noises away: then the pair IS a duplicate)
        if np.array_equal(e[:, a], e[:, b]):
            expect[a, b] = expect[b, a] = True
    assert np.array_equal(nan_want, expect)
    assert np.array_equal(np.isnan(got), nan_want)
    ok = ~nan_want
    # (f32 STORAGE: same f64 arithmetic on the same stored values; the result is rounded to f32 when it is stored)
    tol = 1e-10 if dtype == "float64" else 1.2e-7
    np.testing.assert_allclose(got[ok], want[ok], atol=tol, rtol=0)
    for (a, b), noise in near.items():
        if expect[a, b]:
            continue
        assert abs(got[a, b] - want[a, b]) < tol and abs(got[b, a] - want[b, a]) < tol, (a, b, noise, got[a, b], want[a, b])
    assert abs(got[4, 5] - want[4, 5]) < tol and abs(got[5, 4] - want[5, 4]) < tol
    np.testing.assert_allclose(got[61][ok[61]], want[61][ok[61]], atol=tol, rtol=0)
    # the element-wise kernel (raw f64 moments of A and of d_c): the same pairs, except that it carries the nearly constant d_c of
    # row 61 - variance 1e-11 of its sum of squares, nothing a velocity vector looks like - to 1e-5 only
    okk = ok & ~np.isnan(kern)
    okk[61, :] = False
    np.testing.assert_allclose(kern[okk], want[okk], atol=1e-9 if dtype == "float64" else 2e-4)
    np.testing.assert_allclose(kern[61][ok[61]], want[61][ok[61]], atol=1e-4 if dtype == "float64" else 5e-2)
    # a row block in the middle (tile edges inside the matrix) and the reference's accumulate-into semantics
    blk = ops.coldeltacor_full(E, D, ops.LINEAR, cell0=37, C_out=70).cpu().numpy()
    okb = ok[37:107]
    assert np.array_equal(np.isnan(blk), ~okb) and np.array_equal(blk[okb], got[37:107][okb])
    rm = torch.full((C, C), 2.0, dtype=E.dtype, device=E.t.device)
    acc = ops.coldeltacor_full(E, D, ops.LINEAR, rm=rm, accumulate=True).cpu().numpy()
    assert np.array_equal(np.isnan(acc), nan_want)
    np.testing.assert_allclose(acc[ok], got[ok] + 2.0, atol=1e-6 if dtype == "float32" else 1e-14)


@pytest.mark.parametrize("seed", list(range(10)))
def test_coldeltacor_full_linear_random_degeneracies(ops, oracle, seed):
    """Randomised shapes, row blocks and degeneracies for the matrix-core route with its repair launch: copies of cells at random noise levels
    (1e-2 ... 1e-12, additive and multiplicative), exact duplicates, constant and nearly constant d columns, a random row block (cell0 / C_out:
    the flag words of a block start at the block's first row) - against the oracle's restatement of speedboosted.pyx:13-87, every finite pair to
    1e-10 in f64, NaN pattern equal."""
    rng = np.random.default_rng(1000 + seed)
    G = int(rng.integers(40, 1400))
    C = int(rng.integers(20, 330))
    e, d = rng.gamma(2.0, 1.0, (G, C)) * (rng.random((G, C)) < 0.8), rng.normal(size=(G, C))
    for _ in range(int(rng.integers(1, 9))):
        a, b = rng.choice(C, 2, replace=False)
        kind = rng.integers(0, 4)
        noise = 10.0 ** rng.uniform(-12, -2)
        if kind == 0:
            e[:, b] = e[:, a] + noise * rng.normal(size=G)
        elif kind == 1:
            e[:, b] = e[:, a] * (1.0 + noise)
        elif kind == 2:
            e[:, b] = e[:, a]
        else:
            d[:, b] = rng.choice([0.0, 2.5]) + (noise if rng.random() < 0.5 else 0.0) * rng.normal(size=G)
    want = oracle.coldeltacor(e, d, "linear", 0.0)
    E, D = ops.CellMatrix.from_genes_major(e, "float64"), ops.CellMatrix.from_genes_major(d, "float64")
    got = ops.coldeltacor_full(E, D, ops.LINEAR).cpu().numpy()
    # A nearly constant d column is ill-conditioned in the REFERENCE'S arithmetic as well: b - mean(b) carries the rounding of the mean, eps |mean| against
    # deviations of size std(b) - two correct centred evaluations (other summation orders) differ by about eps |mean| / std.  The bar scales with it; a
    # column that is constant up to 1e-13 of its size is noise in both and is skipped.
    sd, mx = d.std(0), np.abs(d).max(0)
    okrow = ~((sd > 0) & (sd <= 1e-13 * np.maximum(1.0, mx)))
    assert np.array_equal(np.isnan(got)[okrow], np.isnan(want)[okrow])
    tol = 1e-10 + 16 * np.finfo(np.float64).eps * mx / np.where(sd > 0, sd, 1.0)
    fin = np.isfinite(want) & okrow[:, None]
    assert np.all(np.abs(got - want)[fin] <= np.broadcast_to(tol[:, None], got.shape)[fin]), float(np.max((np.abs(got - want) / tol[:, None])[fin]))
    c0 = int(rng.integers(0, C - 1))
    n = int(rng.integers(1, C - c0 + 1))
    blk = ops.coldeltacor_full(E, D, ops.LINEAR, cell0=c0, C_out=n).cpu().numpy()
    assert np.array_equal(np.isnan(blk), np.isnan(got[c0:c0 + n]))
    okb = ~np.isnan(blk)
    np.testing.assert_allclose(blk[okb], got[c0:c0 + n][okb], atol=1e-12, rtol=0)


def test_coldeltacor_full_linear_pooled_neighbours(ops, oracle):
    """The population the repair pass exists for: kNN-pooled cells (every cell the mean of itself and its neighbours, so neighbours
    share most of their pool) - all pairs to 1e-10 in f64 against the oracle, NaN pattern equal."""
    rng = np.random.default_rng(5)
    G, C, k = 1500, 200, 30
    raw = rng.poisson(rng.gamma(0.5, 2.0, (G, 1)) * rng.gamma(5.0, 0.2, (1, C))).astype(np.float64)
    pos = rng.normal(size=(C, 3))
    nn = np.argsort(((pos[:, None, :] - pos[None, :, :]) ** 2).sum(-1), axis=1)[:, :k + 1]
    e = np.stack([raw[:, nn[c]].mean(1) for c in range(C)], axis=1)
    d = rng.normal(size=(G, C)) * 0.1
    want = oracle.coldeltacor(e, d, "linear", 0.0)
    got = ops.coldeltacor_full(ops.CellMatrix.from_genes_major(e, "float64"), ops.CellMatrix.from_genes_major(d, "float64"), ops.LINEAR).cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want)
    np.testing.assert_allclose(got[ok], want[ok], atol=1e-10, rtol=0)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_coldeltacor_full_linear_padding_and_pitch(ops, oracle, dtype):
    """The matrix-core route contracts over whole 16-gene slabs up to the row pitch: a non-zero (or NaN) padding column is refused
    by ops.coldeltacor_full(validate=True) instead of giving silently wrong correlations, and a pitch that does not hold whole
    slabs goes to the element-wise kernel."""
    rng = np.random.default_rng(3)
    G, C = 100, 40
    e, d = rng.gamma(2.0, 1.0, (G, C)), rng.normal(size=(G, C))
    want = oracle.coldeltacor(e, d, "linear", 0.0)
    off = ~np.eye(C, dtype=bool)
    tdt = getattr(torch, dtype)
    for poison in (7.0, float("nan")):
        E, D = ops.CellMatrix.from_genes_major(e, dtype), ops.CellMatrix.from_genes_major(d, dtype)
        assert E.ld > G
        E.t[3, G + 1] = poison
        with pytest.raises(ValueError, match="padding"):
            ops.coldeltacor_full(E, D, ops.LINEAR)
    # pitch 108: not a multiple of 16 -> element-wise kernel, which reads genes 0 .. G - 1 only (the padding may hold anything)
    t_e = torch.full((C, 108), 5.0, dtype=tdt, device="cuda")
    t_d = torch.full((C, 108), float("nan"), dtype=tdt, device="cuda")
    t_e[:, :G] = torch.from_numpy(e.T.copy()).to(tdt)
    t_d[:, :G] = torch.from_numpy(d.T.copy()).to(tdt)
    got = ops.coldeltacor_full(ops.CellMatrix(t_e, G), ops.CellMatrix(t_d, G), ops.LINEAR).cpu().numpy()
    np.testing.assert_allclose(got[off], want[off], atol=1e-9 if dtype == "float64" else 5e-5)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("G,C", [(1, 2), (2, 3), (17, 65), (33, 129), (2049, 70), (500, 257)])
def test_coldeltacor_full_linear_edge_shapes(ops, oracle, dtype, G, C):
    """vcy_coldeltacor_full_linear on shapes around its tile (128 x 64 cells) and slab (16 / 32 genes) edges, against the oracle's
    speedboosted._colDeltaCor restatement; a single gene gives zero variance everywhere (NaN in the reference as well)."""
    rng = np.random.default_rng(G * 1000 + C)
    e, d = rng.gamma(2.0, 1.0, (G, C)), rng.normal(size=(G, C))
    want = oracle.coldeltacor(e, d, "linear", 0.0)
    E, D = ops.CellMatrix.from_genes_major(e, dtype), ops.CellMatrix.from_genes_major(d, dtype)
    got = ops.coldeltacor_full(E, D, ops.LINEAR).cpu().numpy()
    off = ~np.eye(C, dtype=bool)
    assert np.isnan(got[~off]).all()
    if G < 3:                                                # one or two genes: the centred sums of the reference are zero or +-1 up to rounding
        fin = np.isfinite(want) & np.isfinite(got) & off
        np.testing.assert_allclose(got[fin], want[fin], atol=1e-6)
        return
    ok = np.isfinite(want) & off
    assert np.isfinite(got[ok]).all()
    np.testing.assert_allclose(got[ok], want[ok], atol=1e-10 if dtype == "float64" else 5e-5)


def test_coldeltacor_full_linear_register_staged_form(ops, tmp_path):
    """The fallback form of the linear all-pairs kernel (slabs staged through registers; taken for row pitches that do not hold whole
    128-byte slab rows, or with VCY_NT_DMA=0 - an environment switch the library reads once, hence the subprocess) gives what the
    LDS-DMA form gives, to the rounding of a different order of the contraction."""
    import os
    import subprocess
    import sys
    rng = np.random.default_rng(77)
    G, C = 900, 200
    e, d = rng.gamma(2.0, 1.0, (G, C)), rng.normal(size=(G, C))
    np.savez(tmp_path / "in.npz", e=e, d=d)
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import velocyto_amd; from velocyto_amd import ops; z = np.load(%r); "
            "out = {}; \n"
            "for dt in ('float64', 'float32'):\n"
            "    E, D = ops.CellMatrix.from_genes_major(z['e'], dt), ops.CellMatrix.from_genes_major(z['d'], dt)\n"
            "    out[dt] = ops.coldeltacor_full(E, D, ops.LINEAR).cpu().numpy()\n"
            "np.savez(%r, **out)") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path / "in.npz"), str(tmp_path / "out.npz"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, VCY_NT_DMA="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    staged = np.load(tmp_path / "out.npz")
    for dt, tol in (("float64", 1e-12), ("float32", 1e-12)):          # f32 storage: the same f64 arithmetic on the same stored values
        E, D = ops.CellMatrix.from_genes_major(e, dt), ops.CellMatrix.from_genes_major(d, dt)
        dma = ops.coldeltacor_full(E, D, ops.LINEAR).cpu().numpy()
        assert np.array_equal(np.isnan(dma), np.isnan(staged[dt]))
        ok = ~np.isnan(dma)
        np.testing.assert_allclose(staged[dt][ok].astype(np.float64), dma[ok].astype(np.float64), atol=tol if dt == "float64" else 2e-7)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_coldeltacor_partial_wide_lists_are_tiled(ops, oracle, dtype):
    """nrndm > 256: the grouped kernel walks the list in column tiles (one launch each) on index-sorted rows and the
    wrapper restores the caller's column order; duplicates and unsorted rows included."""
    rng = np.random.default_rng(23)
    G, C, nr = 500, 96, 700
    e, d = rng.gamma(2.0, 1.0, (G, C)) * (rng.random((G, C)) < 0.7), rng.normal(size=(G, C))
    ixs = rng.integers(0, C, (C, nr))
    want = oracle.coldeltacor_partial_compact(e, d, ixs, "sqrt", 1e-10)
    E, D = ops.CellMatrix.from_genes_major(e, dtype), ops.CellMatrix.from_genes_major(d, dtype)
    got = ops.coldeltacor_partial(E, D, ixs, ops.SQRT, ops.RULES_PARTIAL, 1e-10).cpu().numpy()
    ok = ~np.isnan(want)
    assert np.array_equal(np.isnan(got), ~ok)
    np.testing.assert_allclose(got[ok], want[ok], atol=1e-10 if dtype == "float64" else 5e-5)
    srt = np.sort(ixs, axis=1)
    got_sorted = ops.coldeltacor_partial(E, D, srt, ops.SQRT, ops.RULES_PARTIAL, 1e-10).cpu().numpy()
    np.testing.assert_array_equal(np.nan_to_num(np.take_along_axis(got_sorted, np.argsort(np.argsort(ixs, axis=1, kind="stable"), axis=1, kind="stable"), 1), nan=7.0),
                                  np.nan_to_num(got, nan=7.0))


def test_coldeltacor_partial_edge_shapes(ops, oracle):
    rng = np.random.default_rng(5)
    for G, C, nr in ((1, 2, 1), (5, 3, 2), (63, 7, 7), (260, 5, 300)):
        e, d = rng.random((G, C)), rng.normal(size=(G, C))
        ixs = rng.integers(0, C, (C, nr))
        got = ops.coldeltacor_partial(ops.CellMatrix.from_genes_major(e, "float64"), ops.CellMatrix.from_genes_major(d, "float64"),
                                      ixs, ops.SQRT, ops.RULES_PARTIAL, 0.0).cpu().numpy()
        with np.errstate(all="ignore"):
            ref = oracle.coldeltacor_partial_compact(e, d, ixs, "sqrt", 0.0)
        ok = np.isfinite(ref) & (ixs != np.arange(C)[:, None]) & (G > 2)
        np.testing.assert_allclose(got[ok], ref[ok], atol=1e-9)
    with pytest.raises(ValueError):
        ops.coldeltacor_partial(ops.CellMatrix.from_genes_major(e), ops.CellMatrix.from_genes_major(d), np.full((C, 2), C), ops.SQRT)


def test_scatter_rows_duplicates_accumulate(ops):
    vals = torch.tensor([[1.0, 2.0, 4.0]], dtype=torch.float64, device="cuda")
    rm = ops.scatter_rows(vals, np.array([[2, 2, 0]]), 4).cpu().numpy()
    np.testing.assert_array_equal(rm, [[4.0, 0.0, 3.0, 0.0]])


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_knn_pool_golden(ops, golden, dtype):
    from scipy import sparse
    g = golden("neighbors")
    C = g["space"].shape[0]
    w = sparse.csr_matrix((g["w_data"], g["w_indices"], g["w_indptr"]), shape=(C, C))
    data = ops.CellMatrix.from_genes_major(g["data"], dtype)
    for slab in (0, 4, 16):
        out = ops.knn_pool(data, w.indptr, w.indices, w.data, slab_genes=slab).to_genes_major()
        np.testing.assert_allclose(out, g["convolved"], rtol=1e-12 if dtype == "float64" else 2e-6, atol=1e-12 if dtype == "float64" else 1e-6)
    mx = ops.knn_pool(data, w.indptr, w.indices, w.data, maximum=True).to_genes_major()
    np.testing.assert_allclose(mx, np.maximum(g["convolved"], g["data"]), rtol=1e-12 if dtype == "float64" else 2e-6, atol=1e-6)
    part = ops.knn_pool(data, w.indptr[50:121] - w.indptr[50], w.indices[w.indptr[50]:w.indptr[120]], w.data[w.indptr[50]:w.indptr[120]],
                        cell0=50, C_out=70, maximum=True).to_genes_major()
    np.testing.assert_array_equal(part, mx[:, 50:120])


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_pooling_in_canonical_row_order_keeps_equal_neighbourhoods_equal(ops, oracle, dtype):
    """Two cells with the SAME closed neighbourhood (mutual neighbours in a tight cluster) pool the same cells.  The reference
    pools through scipy with column-sorted rows (analysis.py:1006-1013), so their pooled vectors are bitwise equal, every
    difference is exactly zero, the zero rule of speedboosted.pyx:372 applies and the pair's correlation is NaN (mapped to 1 at
    analysis.py:1605).  ops.canonical_graph_rows puts device-built graph rows in that order: same NaN pattern and the same
    values as the oracle for the whole path; nearest-first rows leave +-1e-16 relative noise that the partial-sqrt transform
    lifts to +-sqrt(psc) per gene - a finite correlation where the reference has none."""
    rng = np.random.default_rng(42)
    C, G, k, P = 120, 1500, 5, 4
    space = rng.normal(size=(C, P)) * 10.0
    space[:6] = 100.0 + rng.normal(size=(6, P)) * 1e-3             # a cluster of k + 1 cells far from the rest: one closed neighbourhood
    S = rng.poisson(rng.gamma(0.5, 2.0, (G, 1)) * np.ones((1, C))).astype(np.float64)
    U = rng.poisson(0.3 * rng.gamma(0.5, 2.0, (G, 1)) * np.ones((1, C))).astype(np.float64)
    fS = S.sum(0).mean() / S.sum(0)
    S_sz, U_sz = S * fS[None, :], U * fS[None, :]
    _, w, Sx_o, Ux_o = oracle.knn_imputation(S_sz, U_sz, space, k=k)
    assert np.array_equal(Sx_o[:, 0], Sx_o[:, 3])                   # the reference's order: bitwise equal pooled vectors
    gam = oracle.fit_slope(Ux_o, Sx_o)
    gam = np.where(np.isfinite(gam), gam, 0.0)
    _, _, dS, _ = oracle.velocity_chain(Sx_o, Ux_o, gam, None)
    d_o = oracle.delta_transform(Sx_o, Sx_o + dS, "sqrt", 1e-10)
    ixs = np.stack([rng.choice(C, 16, replace=False) for _ in range(C)])
    ixs[0, :5] = [1, 2, 3, 4, 5]
    want = oracle.coldeltacor_partial_compact(Sx_o, d_o, ixs, "sqrt", 1e-10)
    assert np.isnan(want[0, :5]).all()
    # the device path with device-built rows [self | nearest first] -> canonical order
    dev = ops.require_gpu()
    tdt = torch.float64 if dtype == "float64" else torch.float32
    idx, dist = ops.knn_search(space, k)
    wrow = torch.cat([torch.ones((C, 1), device=dev, dtype=tdt), (dist > 0).to(tdt)], 1)
    wrow = wrow / wrow.sum(1, keepdim=True)
    rows = torch.cat([torch.arange(C, device=dev, dtype=torch.int32)[:, None], idx], 1)
    indptr = torch.arange(0, (C + 1) * (k + 1), k + 1, device=dev, dtype=torch.int64)
    cS, cU = ops.CountMatrix.from_genes_major(S.astype(np.uint16)), ops.CountMatrix.from_genes_major(U.astype(np.uint16))

    def path(ind, ww):
        Sx, Ux = ops.knn_pool_counts(cS, cU, fS, fS, indptr, ind.contiguous(), ww.contiguous(), dtype=dtype)
        g = ops.fit_slope(Ux, Sx)
        g[~torch.isfinite(g)] = 0.0
        return Sx, ops.coldeltacor_partial_fused(Sx, Ux, g, None, ixs, ops.SQRT, ops.RULES_PARTIAL, 1e-10).cpu().numpy()
    Sx, got = path(*ops.canonical_graph_rows(rows, wrow))
    assert torch.equal(Sx.t[0], Sx.t[3])
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = np.isfinite(want)
    np.testing.assert_allclose(got[ok], want[ok], atol=CORR_ATOL[dtype])
    _, raw = path(rows, wrow)                                       # nearest-first rows: the same pairs come out finite
    assert np.isfinite(raw[0, :5]).any()


@pytest.mark.parametrize("narrow", [True, False])
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_knn_pool_counts(ops, oracle, golden, dtype, narrow):
    """Pooling gathered from the uint16 count layers + per-cell size factors == pooling of the normalised floats."""
    from scipy import sparse
    g = golden("pipeline")
    S, U = g["S"], g["U"]                                   # uint16 (G, C) like a loom
    assert ops.CountMatrix.representable(S) and not ops.CountMatrix.representable(S.astype(float))
    C = S.shape[1]
    fS, fU = S.sum(0).mean() / S.sum(0), U.sum(0).mean() / U.sum(0)
    knn = oracle.knn_graph(g["pcs"][:, :10], 12)
    w = oracle.connectivity_to_weights(knn)
    w.sort_indices()
    cS, cU = ops.CountMatrix.from_genes_major(S, narrow=narrow), ops.CountMatrix.from_genes_major(U, narrow=narrow)
    assert cS.t.dtype == ((torch.uint8 if S.max() <= 255 else torch.int16) if narrow else torch.int16)
    assert cS.ld % 64 == 0 and int(cS.t[:, cS.G:].to(torch.int32).abs().sum()) == 0
    if narrow:       # narrowing after the fact, and a wide + narrow pair in one call
        wide = ops.CountMatrix.from_genes_major(np.minimum(S, 255), narrow=False)
        assert wide.t.dtype == torch.int16 and wide.narrowed().t.dtype == torch.uint8 and torch.equal(wide.narrowed().as_int32(), wide.as_int32())
        over = S.copy(); over[0, 0] = 300
        assert ops.CountMatrix.from_genes_major(over).t.dtype == torch.int16 and ops.CountMatrix.from_genes_major(over).narrowed().t.dtype == torch.int16
        mixS, mixU = ops.knn_pool_counts(ops.CountMatrix.from_genes_major(S, narrow=False), cU, fS, fU, w.indptr, w.indices, w.data, dtype=dtype)
        np.testing.assert_allclose(mixS.to_genes_major(), g["Sx"], rtol=1e-12 if dtype == "float64" else 3e-6, atol=1e-12 if dtype == "float64" else 3e-6)
    np.testing.assert_array_equal(cS.to_float("float64").to_genes_major(), S.astype(float))
    rt = 1e-12 if dtype == "float64" else 3e-6
    for slab in (0, 8, 24):
        Sx, Ux = ops.knn_pool_counts(cS, cU, fS, fU, w.indptr, w.indices, w.data, dtype=dtype, slab_genes=slab)
        np.testing.assert_allclose(Sx.to_genes_major(), g["Sx"], rtol=rt, atol=rt)
        np.testing.assert_allclose(Ux.to_genes_major(), g["Ux"], rtol=rt, atol=rt)
        assert float(Sx.t[:, Sx.G:].abs().sum()) == 0.0
    Sx1 = ops.knn_pool_counts(cS, None, fS, None, w.indptr, w.indices, w.data, dtype=dtype)
    np.testing.assert_allclose(Sx1.to_genes_major(), g["Sx"], rtol=rt, atol=rt)
    # maximum=True with diag=2 (golden max_*) and a row block with an order
    conn = (knn > 0).astype(float).tolil()
    conn.setdiag(2.0)
    w2 = sparse.csr_matrix(oracle.connectivity_to_weights(conn.tocsr(), diag=2.0))
    w2.sort_indices()
    Sx, Ux = ops.knn_pool_counts(cS, cU, fS, fU, w2.indptr, w2.indices, w2.data, dtype=dtype, maximum=True)
    np.testing.assert_allclose(Sx.to_genes_major(), g["max_Sx"], rtol=rt, atol=rt)
    np.testing.assert_allclose(Ux.to_genes_major(), g["max_Ux"], rtol=rt, atol=rt)
    lo, hi = 40, 123
    part = ops.knn_pool_counts(cS, None, fS, None, w2.indptr[lo:hi + 1] - w2.indptr[lo], w2.indices[w2.indptr[lo]:w2.indptr[hi]],
                               w2.data[w2.indptr[lo]:w2.indptr[hi]], dtype=dtype, maximum=True, cell0=lo, C_out=hi - lo,
                               order=torch.randperm(hi - lo).to(torch.int32))
    np.testing.assert_array_equal(part.to_genes_major(), Sx.to_genes_major()[:, lo:hi])
    big = np.array([[70000, 1]], dtype=np.int64)
    assert not ops.CountMatrix.representable(big)


@pytest.mark.parametrize("include_self", [False, True])
def test_knn_search_vs_oracle(ops, oracle, golden, include_self):
    g = golden("neighbors")
    space = g["space"]
    for k in (1, 9, 41, 200):
        idx, dist = ops.knn_search(space, k, include_self=include_self)
        od, oi = oracle.knn_search(space, k, include_self=include_self)
        idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
        np.testing.assert_allclose(dist, od, atol=1e-12)
        assert np.array_equal(idx, oi)                     # incl. the duplicated cell 10/11: ties by index
    if not include_self:
        o = np.argsort(oi := oracle.knn_search(space, 9)[1], axis=1)
        diff = np.take_along_axis(ops.knn_search(space, 9)[0].cpu().numpy(), o, 1) != g["knn_indices"]
        assert diff.sum() <= 4                             # reference (sklearn) may order exact ties differently


def test_knn_search_blocks_and_limits(ops, oracle):
    rng = np.random.default_rng(3)
    space = rng.normal(size=(1500, 30))
    idx, dist = ops.knn_search(space, 30, query_block=512)
    od, oi = oracle.knn_search(space, 30)
    assert np.array_equal(idx.cpu().numpy(), oi)
    np.testing.assert_allclose(dist.cpu().numpy(), od, atol=1e-12)
    i2, d2 = ops.knn_search(space, 30, q0=700, Q=33)
    assert np.array_equal(i2.cpu().numpy(), oi[700:733])
    # large k (candidate lists sorted in the global workspace instead of LDS), e.g. n_neighbors = C/5 defaults
    emb = rng.normal(size=(6000, 2))
    i3, d3 = ops.knn_search(emb, 5000, q0=100, Q=19)
    od, oi = oracle.knn_search(emb, 5000)
    assert np.array_equal(i3.cpu().numpy(), oi[100:119])
    np.testing.assert_allclose(d3.cpu().numpy(), od[100:119], atol=1e-12)
    with pytest.raises(ValueError):
        ops.knn_search(emb, 6000)


def test_knn_search_rowfree_fallbacks(ops, oracle):
    """The row-free fast path (k + 8 <= 128): a thread tracks four candidates of its strided slice (j = t mod 256).
    (a) six near neighbours in ONE slice -> that thread overflows and the workgroup rescans its slice;
    (b) hundreds of exact duplicates -> more than eight threads overflow -> the query recomputes its row and takes the
        radix-select path; ties must still come out in index order.  Both against the oracle, bit-exact."""
    rng = np.random.default_rng(11)
    C, P, k = 2600, 6, 12
    space = rng.normal(size=(C, P)) * 5.0
    q = 1000
    space[300:900] = space[300]                             # 600 identical points
    space[2000] = space[300] + 1e-4
    planted = [7, 263, 1031, 1287, 1543, 1799, 2055]        # all in the slice of thread 7
    for n, j in enumerate(planted):
        space[j] = space[q] + 1e-3 * (n + 1) * np.ones(P)
    idx, dist = ops.knn_search(space, k)
    od, oi = oracle.knn_search(space, k)
    assert np.array_equal(idx.cpu().numpy(), oi)
    np.testing.assert_allclose(dist.cpu().numpy(), od, atol=1e-12)
    assert set(oi[q][:7]) == set(planted) and (oi[2000] < 900).all() and (oi[2000] >= 300).all()
    i2, d2 = ops.knn_search(space, k, include_self=True)
    od2, oi2 = oracle.knn_search(space, k, include_self=True)
    assert np.array_equal(i2.cpu().numpy(), oi2)
    # external queries through the same path
    qs = space[[q, 2000, 5]] + 1e-5
    i3, d3 = ops.knn_query(space, qs, k)
    od3, oi3 = oracle.knn_query(space, qs, k)
    assert np.array_equal(i3.cpu().numpy(), oi3)
    np.testing.assert_allclose(d3.cpu().numpy(), od3, atol=1e-12)


def test_fit_slope_golden(ops, golden):
    g = golden("fits")
    for dtype, rtol in (("float64", 2e-7), ("float32", 1e-5)):
        Y, X = ops.CellMatrix.from_genes_major(g["Y"], dtype), ops.CellMatrix.from_genes_major(g["X"], dtype)
        got = ops.fit_slope(Y, X).cpu().numpy()
        assert got.dtype == np.float32 and np.isnan(got[0]) and got[1] == 0
        np.testing.assert_allclose(got[1:], g["fit_slope"][1:], rtol=rtol)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_gene_quantiles(ops, dtype):
    rng = np.random.default_rng(9)
    for G, C in ((7, 5), (40, 1000), (130, 3001)):
        a = rng.gamma(1.0, 2.0, (G, C)) * (rng.random((G, C)) < 0.5)     # many exact ties at 0
        a[0] = 0
        a[1] = -a[1]
        m = ops.CellMatrix.from_genes_major(a, dtype)
        qs = [0, 2, 50, 98, 99.9, 100]
        got = ops.gene_quantiles(m, qs).cpu().numpy()
        ref = np.percentile(m.to_genes_major(), qs, axis=1)              # of the stored (possibly f32-rounded) values
        np.testing.assert_allclose(got, ref, rtol=1e-14, atol=0)
        b = rng.gamma(1.0, 1.0, (G, C))
        sa, sb = rng.random(G) + 0.5, rng.random(G) + 0.5
        m2 = ops.CellMatrix.from_genes_major(b, dtype)
        got = ops.gene_quantiles(m, [2, 98], M2=m2, scale_a=torch.from_numpy(sa).cuda(), scale_b=torch.from_numpy(sb).cuda()).cpu().numpy()
        np_t = np.float64 if dtype == "float64" else np.float32
        Z = (m.to_genes_major(np_t) / sa.astype(np_t)[:, None] + m2.to_genes_major(np_t) / sb.astype(np_t)[:, None])
        np.testing.assert_allclose(got, np.percentile(Z.astype(np.float64), [2, 98], axis=1), rtol=1e-14, atol=0)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_pool_two_weight_sets_equals_two_poolings(ops, dtype):
    """vcy_knn_pool_w2 (calculate_embedding_shift's real + control products in one gather) against two vcy_knn_pool launches:
    the same fma chain per output, so the same bits; ragged rows, an empty row, a cell block, a schedule."""
    rng = np.random.default_rng(12)
    C, G = 300, 1003
    data = ops.CellMatrix.from_genes_major(rng.gamma(1.0, 1.0, (G, C)), dtype)
    lens = rng.integers(0, 23, C)
    lens[5] = 0
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    indices = rng.integers(0, C, indptr[-1]).astype(np.int32)
    w, w2 = rng.normal(size=indptr[-1]), rng.normal(size=indptr[-1])
    order = torch.from_numpy(rng.permutation(C).astype(np.int32)).cuda()
    a, b = ops.knn_pool_w2(data, indptr, indices, w, w2, order=order)
    ra, rb = ops.knn_pool(data, indptr, indices, w), ops.knn_pool(data, indptr, indices, w2)
    assert torch.equal(a.t, ra.t) and torch.equal(b.t, rb.t)
    c0, n = 40, 100                                                       # rows of a block of output cells
    ip = (indptr[c0:c0 + n + 1] - indptr[c0]).astype(np.int64)
    sl = slice(indptr[c0], indptr[c0 + n])
    a, b = ops.knn_pool_w2(data, ip, indices[sl], w[sl], w2[sl], cell0=c0, C_out=n)
    assert torch.equal(a.t, ra.t[c0:c0 + n]) and torch.equal(b.t, rb.t[c0:c0 + n])


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_permute_rows_nsign_is_a_uniform_independent_shuffle_per_gene(ops, dtype):
    """analysis.py:2407-2420 (numba: np.random.shuffle of every gene's row, then random signs).  The device version evaluates a
    keyed permutation per gene: every gene's output is a permutation of its input up to signs; permutations are reproducible from
    the seed, differ between genes and seeds, and behave like uniform draws (where a cell lands, fixed points, no memory of the
    source position, balanced signs independent of the shuffle)."""
    rng = np.random.default_rng(4)
    for C, G in ((1, 3), (2, 70), (17, 70), (1000, 300), (5000, 70)):
        a = np.stack([rng.permutation(C) + 1.0 for _ in range(G)])          # distinct positive values per gene: the value names its source cell
        m = ops.CellMatrix.from_genes_major(a, dtype)
        out = ops.permute_rows_nsign(m, 15071990, gene_major=False)
        assert torch.all(out.t[:, G:] == 0)                                 # padding columns of the layout
        assert torch.equal(ops.permute_rows_nsign(m, 15071990, gene_major=True).t, out.t)    # the gene-major route: the same shuffle
        o = out.to_genes_major()
        np.testing.assert_array_equal(np.sort(np.abs(o), 1), np.sort(a, 1))
        np.testing.assert_array_equal(ops.permute_rows_nsign(m, 15071990).to_genes_major(), o)
        if C >= 17:
            o2 = ops.permute_rows_nsign(m, 15071991).to_genes_major()
            assert (np.abs(o2) != np.abs(o)).mean() > 0.8
            src = np.abs(o) - 1                                             # rank of the value that landed in each cell
            inv = np.argsort(a, 1)                                          # ... back to its source cell
            pi = np.take_along_axis(inv, src.astype(np.int64), 1)
            assert (pi[0] != pi[1]).mean() > 0.8                            # genes are shuffled independently
            assert (pi == np.arange(C)[None, :]).sum(1).mean() < 3.0        # ~1 fixed point per permutation
            if C >= 1000:
                r = np.array([np.corrcoef(pi[g], np.arange(C))[0, 1] for g in range(G)])
                assert np.abs(r).max() < 6 / np.sqrt(C) and abs(r.mean()) < 4 / np.sqrt(C * G)
                d = (pi - np.arange(C)[None, :]) % C                        # displacement: uniform over [0, C)
                h = np.bincount((d.ravel() * 20 // C).astype(np.int64), minlength=20)
                e = d.size / 20
                assert ((h - e) ** 2 / e).sum() < 60                        # chi-square, 19 degrees of freedom
            sg = np.sign(o)
            assert abs(sg.mean()) < 5 / np.sqrt(sg.size)
            assert abs(np.corrcoef(sg.ravel()[:-1], sg.ravel()[1:])[0, 1]) < 5 / np.sqrt(sg.size)
    # where cell 0 comes from, over many genes: uniform over a small population
    C, G = 50, 20000
    a = np.tile(np.arange(1.0, C + 1), (G, 1))
    o = np.abs(ops.permute_rows_nsign(ops.CellMatrix.from_genes_major(a, dtype), 7).to_genes_major())
    for cell in (0, 49):
        h = np.bincount(o[:, cell].astype(np.int64) - 1, minlength=C)
        assert ((h - G / C) ** 2 / (G / C)).sum() < 120                    # chi-square, 49 degrees of freedom


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_gene_quantiles_every_register_variant(ops, dtype):
    """The selection with the gene's keys in registers (fit.hip k_gene_quantiles_reg, one instantiation per band of cell counts) and
    the row-re-reading kernel above 65 536 (f32) / 32 768 (f64) cells: exact order statistics with numpy's interpolation, plain and
    restricted to the cells above / at-or-below a per-gene threshold of a second matrix."""
    rng = np.random.default_rng(91)
    G = 5
    for C in (1025, 8192, 9000, 20000, 24577, 33000, 49153, 50000, 57345, 65536, 70001):
        a = rng.gamma(1.0, 2.0, (G, C)) * (rng.random((G, C)) < 0.6)
        a[1] = -a[1]
        a[2] = np.round(a[2])                                              # heavy ties
        m = ops.CellMatrix.from_genes_major(a, dtype)
        stored = m.to_genes_major()
        qs = [0, 2, 37.5, 98, 99.99, 100]
        got = ops.gene_quantiles(m, qs).cpu().numpy()
        np.testing.assert_allclose(got, np.percentile(stored, qs, axis=1), rtol=1e-14, atol=0)
        src = rng.random((G, C))
        thr = np.array([0.5, 0.9, 0.001, 2.0, -1.0])                       # gene 3: no cell above; gene 4: every cell above
        msrc = ops.CellMatrix.from_genes_major(src, dtype)
        sst = msrc.to_genes_major()
        for mode in (1, 2):
            got = ops.gene_quantiles(m, [2, 98], mask_src=msrc, mask_thr=torch.from_numpy(thr).cuda(), mask_mode=mode).cpu().numpy()
            for g in range(G):
                sel = sst[g] > thr[g] if mode == 1 else sst[g] <= thr[g]
                if sel.any():
                    np.testing.assert_allclose(got[:, g], np.percentile(stored[g][sel], [2, 98]), rtol=1e-14, atol=0)
                else:
                    assert np.isnan(got[:, g]).all()


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_fit_weighted_vs_oracle_exact(ops, oracle, golden, fit_parity, dtype):
    g = golden("fits")
    rt = 1e-9 if dtype == "float64" else 2e-5
    Y, X, W = (ops.CellMatrix.from_genes_major(g[k], dtype) for k in ("Y", "X", "W"))
    m, q, r2 = ops.fit_weighted(Y, X, 0, W=W)
    me, qe, r2e = oracle.fit_slope_weighted_offset(g["Y"], g["X"], g["W"], exact=True)
    assert np.isnan(m.cpu().numpy()[0])
    np.testing.assert_allclose(m.cpu().numpy()[1:], me[1:], rtol=rt, atol=rt)
    np.testing.assert_allclose(q.cpu().numpy(), qe, rtol=rt, atol=rt)
    np.testing.assert_allclose(r2.cpu().numpy()[2:], r2e[2:], rtol=10 * rt, atol=10 * rt)
    # vs the reference's L-BFGS-B stopping point (golden), SURVEY.md section 7's bar: rtol 1e-4 on >= 99 % of the genes (here: all of them),
    # the objective never worse (f32 storage: the objective is evaluated on the f64 inputs, the parameters come from their f32 roundings)
    fit_parity(m.cpu().numpy(), q.cpu().numpy(), g["woffset_nolg_m"], g["woffset_nolg_q"], g["Y"], g["X"], g["W"],
               slack=1e-9 if dtype == "float64" else 1e-5, skip=(0,))
    # gamma only, bounded (fit_slope_weighted) and unweighted with intercept (fit_slope_offset)
    m, _, r2 = ops.fit_weighted(Y, X, 0, W=W, fit_offset=False, lo_gamma=0.0)
    me, r2e = oracle.fit_slope_weighted(g["Y"], g["X"], g["W"], exact=True)
    np.testing.assert_allclose(m.cpu().numpy()[1:], me[1:], rtol=rt, atol=rt)
    np.testing.assert_allclose(m.cpu().numpy()[1:], g["weighted_nolg_m"][1:], rtol=1e-4, atol=2e-5)
    m, q, _ = ops.fit_weighted(Y, X, 2, box_q=False)
    np.testing.assert_allclose(m.cpu().numpy()[2:], g["offset_m"][2:], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(q.cpu().numpy()[2:], g["offset_q"][2:], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_velocity_chain_golden(ops, golden, dtype):
    g = golden("pipeline")
    Sx, Ux = ops.CellMatrix.from_genes_major(g["Sx"], dtype), ops.CellMatrix.from_genes_major(g["Ux"], dtype)
    gam, q = torch.from_numpy(g["gammas"]), torch.from_numpy(g["q"])
    out = ops.velocity_chain(Sx, Ux, gam, q, want=("Upred", "velocity", "delta_S", "Sx_sz_t", "dmat"), transform=ops.SQRT, psc=1e-10)
    rt, at = (1e-13, 1e-13) if dtype == "float64" else (3e-6, 3e-6)
    for name in ("Upred", "velocity", "delta_S", "Sx_sz_t"):
        np.testing.assert_allclose(out[name].to_genes_major(), g[name], rtol=rt, atol=at)
    import oracle
    dref = oracle.delta_transform(g["Sx"], g["Sx"] + 1.0 * g["delta_S"], "sqrt", 1e-10)
    np.testing.assert_allclose(out["dmat"].to_genes_major(), dref, rtol=rt, atol=1e-7 if dtype == "float64" else 2e-3)
    out = ops.velocity_chain(Sx, Ux, gam, q, want=("delta_S",), dt_shift=0.7, assumption=1)
    ref = g["delta_S_cu"]
    ok = np.isfinite(ref)
    # exp(-gamma*dt) is a float32 quantity in the reference (numpy float32 exp vs device expf: 1 ulp apart, amplified by (1 - egt)/gamma for small gamma)
    np.testing.assert_allclose(out["delta_S"].to_genes_major()[ok], ref[ok], rtol=1e-5 if dtype == "float64" else 2e-4, atol=1e-5 if dtype == "float64" else 1e-4)


@pytest.mark.parametrize("narrow", [True, False])
@pytest.mark.parametrize("shape", [(1, 1), (5, 17), (33, 64), (70, 1000), (40, 1025)])
def test_knn_pool_counts_ragged_graphs(ops, shape, narrow):
    """Count-layer pooling on ragged CSR graphs (0..9 neighbours per cell, empty rows, repeated neighbours), gene counts
    around the 16-byte / slab boundaries, both storage widths, against a dense numpy product."""
    C, G = shape
    rng = np.random.default_rng(C * 1000 + G)
    S = rng.poisson(2.0, (G, C)).astype(np.uint16)
    if not narrow:
        S[0, 0] = 40000                                      # forces uint16 storage and exercises values above 32767
    deg = rng.integers(0, 10, C)
    indptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    indices = rng.integers(0, C, indptr[-1]).astype(np.int32)
    w = rng.random(indptr[-1])
    scale = rng.random(C) + 0.5
    W = np.zeros((C, C))
    for c in range(C):
        for p in range(indptr[c], indptr[c + 1]):
            W[c, indices[p]] += w[p]
    want = (W * scale[None, :]) @ S.T.astype(np.float64)     # (C, G)
    cS = ops.CountMatrix.from_genes_major(S)
    assert cS.t.dtype == (torch.uint8 if narrow else torch.int16)
    for dtype, tol in (("float64", 1e-12), ("float32", 3e-6)):
        for slab in (0, 16, 1024):
            got = ops.knn_pool_counts(cS, None, scale, None, indptr, indices, w, dtype=dtype, slab_genes=slab)
            np.testing.assert_allclose(got.to_cells_major(), want, rtol=tol, atol=tol * max(1.0, np.abs(want).max()))
            assert float(got.t[:, G:].abs().sum()) == 0.0
    mx = ops.knn_pool_counts(cS, None, scale, None, indptr, indices, w, dtype="float64", maximum=True).to_cells_major()
    np.testing.assert_allclose(mx, np.maximum(want, S.T.astype(np.float64) * scale[:, None]), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("C,P,k", [(2, 1, 1), (9, 3, 8), (257, 2, 5), (300, 7, 100), (300, 7, 121), (513, 1, 30), (1000, 33, 64)])
def test_knn_search_small_and_boundary_shapes(ops, oracle, C, P, k):
    """Both selection kernels around their limits: k + 8 <= 128 takes the row-free path, larger k the row-materialising one;
    candidate counts near the slice width (256), k close to the number of candidates, one feature, duplicated points."""
    rng = np.random.default_rng(C * 7 + P * 3 + k)
    X = rng.normal(size=(C, P))
    if P < 8:
        X = np.round(X, 1)    # coarse grid: plenty of exact distance ties (sequential sums of rounded squares, as numpy's for < 8 terms)
    else:
        X[C // 2] = X[C // 3]                                # exact duplicates tie in any summation order
    for include_self in (False, True):
        if not include_self and k > C - 1:
            continue
        idx, dist = ops.knn_search(X, k, include_self=include_self)
        od, oi = oracle.knn_search(X, k, include_self=include_self)
        assert np.array_equal(idx.cpu().numpy(), oi)
        np.testing.assert_allclose(dist.cpu().numpy(), od, atol=1e-12)


@pytest.mark.parametrize("G,C,nr", [(1535, 33, 8), (1536, 32, 9), (1537, 40, 7), (3073, 31, 12), (100, 64, 255), (64, 70, 257), (6, 35, 16)])
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_coldeltacor_partial_around_kernel_thresholds(ops, oracle, G, C, nr, dtype):
    """Gene counts around the 1536-gene chunk (ragged tails), cell counts around the 32-cell / 8-neighbour limits of the
    grouped kernel, list widths around one 256-column tile - plain and with a schedule over a subset of the cells."""
    rng = np.random.default_rng(G + 31 * C + 7 * nr)
    e, d = rng.gamma(2.0, 1.0, (G, C)) * (rng.random((G, C)) < 0.8), rng.normal(size=(G, C))
    ixs = rng.integers(0, C, (C, nr))
    want = oracle.coldeltacor_partial_compact(e, d, ixs, "sqrt", 1e-10)
    E, D = ops.CellMatrix.from_genes_major(e, dtype), ops.CellMatrix.from_genes_major(d, dtype)
    got = ops.coldeltacor_partial(E, D, ixs, ops.SQRT, ops.RULES_PARTIAL, 1e-10).cpu().numpy()
    ok = np.isfinite(want)
    tol = 1e-10 if dtype == "float64" else 1e-4
    np.testing.assert_allclose(got[ok], want[ok], atol=tol)
    assert np.isnan(got[~ok]).all()
    # schedule over a subset: only those rows are written
    sub = torch.from_numpy(rng.permutation(C)[: max(1, C // 3)].astype(np.int32))
    out = torch.full((C, nr), 5.0, dtype=E.dtype, device=E.t.device)
    ops.coldeltacor_partial(E, D, ixs, ops.SQRT, ops.RULES_PARTIAL, 1e-10, order=sub, out=out)
    out = out.cpu().numpy()
    rows = np.zeros(C, dtype=bool)
    rows[sub.numpy()] = True
    assert (out[~rows] == 5.0).all()
    sel = ok & rows[:, None]
    np.testing.assert_allclose(out[sel], want[sel], atol=tol)


@pytest.mark.parametrize("G,C", [(1, 1), (3, 2), (255, 3), (257, 70), (1030, 257)])
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_stage_b_c_kernels_on_odd_shapes(ops, oracle, G, C, dtype):
    """fit_slope, the per-gene moments, row sums and the fused velocity chain on shapes that are not multiples of the
    vector width, the 256-gene blocks or the cell blocks - against numpy / the oracle."""
    rng = np.random.default_rng(G * 13 + C)
    Sx = rng.gamma(2.0, 1.0, (G, C)) * (rng.random((G, C)) < 0.8)
    Ux = rng.gamma(1.0, 1.0, (G, C)) * (rng.random((G, C)) < 0.7)
    S, U = ops.CellMatrix.from_genes_major(Sx, dtype), ops.CellMatrix.from_genes_major(Ux, dtype)
    Sh, Uh = S.to_genes_major(), U.to_genes_major()              # the stored (possibly f32-rounded) values
    rt = 1e-12 if dtype == "float64" else 2e-6
    gam = ops.fit_slope(U, S).cpu().numpy()
    with np.errstate(all="ignore"):
        ref = np.maximum(0, (Sh * Uh).sum(1) / (Sh * Sh).sum(1))
    ok = np.isfinite(ref)
    np.testing.assert_allclose(gam[ok], ref[ok], rtol=max(rt, 2e-7), atol=1e-12)
    assert np.isnan(gam[~ok]).all()
    mom = ops.gene_moments(U, S).cpu().numpy()
    for got, want in zip(mom, (Sh.sum(1), Uh.sum(1), (Sh * Sh).sum(1), (Sh * Uh).sum(1), (Uh * Uh).sum(1))):
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(ops.row_sums(S).cpu().numpy(), Sh.sum(0), rtol=1e-12, atol=1e-12)
    g32 = np.where(ok, gam, 0.25).astype(np.float32)
    out = ops.velocity_chain(S, U, torch.from_numpy(g32), None, want=("Upred", "velocity", "delta_S", "Sx_sz_t", "dmat"), transform=ops.SQRT, psc=1e-10)
    Upred, vel, dS, Sxt = oracle.velocity_chain(Sh, Uh, g32, None)
    for name, want in (("Upred", Upred), ("velocity", vel), ("delta_S", dS), ("Sx_sz_t", Sxt)):
        np.testing.assert_allclose(out[name].to_genes_major(), want, rtol=max(rt, 1e-6 if dtype == "float32" else rt), atol=1e-6 if dtype == "float32" else 1e-12)
    if dtype == "float64":
        np.testing.assert_allclose(out["dmat"].to_genes_major(), oracle.delta_transform(Sh, Sh + dS, "sqrt", 1e-10), rtol=1e-9, atol=1e-12)
    assert float(out["dmat"].t[:, G:].abs().sum()) == 0.0


@pytest.mark.parametrize("C,n,edim", [(2, 1, 2), (9, 4, 2), (65, 7, 3), (300, 33, 2), (257, 256, 2)])
def test_stage_e_f_kernels_on_odd_shapes(ops, oracle, C, n, edim):
    """transition_prob / delta_embedding in neighbour-list form, the dense Markov matrix and the diffusion steps on
    shapes around the wave and block sizes, against the oracle's dense restatement of analysis.py:1670-1733, 1818-1887."""
    rng = np.random.default_rng(C * 17 + n)
    corr = rng.uniform(-0.5, 0.5, (C, n))
    ixs = np.stack([rng.choice(np.delete(np.arange(C), c), n, replace=False) for c in range(C)])
    emb = rng.normal(size=(C, edim))
    dense = np.zeros((C, C))
    dense[np.arange(C)[:, None], ixs] = corr
    tp_ref, de_ref, _ = oracle.calculate_embedding_shift(dense, ixs, emb, sigma_corr=0.07, expression_scaling=False)
    tp, wd, de = ops.transition_prob(torch.from_numpy(corr).cuda(), ixs, emb, 0.07)
    got = np.zeros((C, C))
    got[np.arange(C)[:, None], ixs] = tp.cpu().numpy()
    np.testing.assert_allclose(got, tp_ref, rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(de.cpu().numpy(), de_ref, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(wd.cpu().numpy(), tp.cpu().numpy() - 1.0 / n, rtol=1e-12, atol=1e-15)
    if edim == 2:
        from scipy import sparse
        P = sparse.csr_matrix(got)
        P.sort_indices()
        for direction in ("forward", "backwards"):
            Pd = P if direction == "forward" else sparse.csr_matrix(P.T)
            Pd.sort_indices()
            tr = ops.prepare_markov(Pd.indptr, Pd.indices, Pd.data, emb, 0.8, 1.7).cpu().numpy()
            ref = oracle.prepare_markov(tp_ref, emb, 0.8, 1.7, direction)
            np.testing.assert_allclose(tr, ref, rtol=1e-11, atol=1e-16)
        x0 = rng.random(C)
        for acc, mode in ((False, "time_evolution"), (True, "path_integral")):
            xf, xa = ops.diffuse(x0 / x0.sum(), torch.from_numpy(ref).cuda(), 9, accumulate=acc)
            want = oracle.diffuse(x0, ref, 9, mode).ravel()
            np.testing.assert_allclose((xa if acc else xf).cpu().numpy(), want, rtol=1e-11)


@pytest.mark.parametrize("n", [1024, 1026, 1301, 4100])
@pytest.mark.parametrize("tdtype", ["float32", "float64"])
def test_diffuse_dense_wide_loads(ops, oracle, n, tdtype):
    """The dense Markov step at sizes that take the 16-byte-load kernel (n a multiple of 4 / 2 columns per thread, aligned base)
    and at sizes / alignments that take the one-column kernel: both against numpy, and bit-identical to each other."""
    rng = np.random.default_rng(n)
    tr = rng.random((n, n)) ** 8
    tr /= tr.sum(1, keepdims=True)
    tr = tr.astype(tdtype)
    x0 = rng.random(n)
    x0 /= x0.sum()
    dev = ops.require_gpu()
    aligned = torch.from_numpy(tr).to(dev)
    buf = torch.empty(n * n + 1, dtype=aligned.dtype, device=dev)
    shifted = buf[1:].view(n, n)                                     # same values, base address off by one element: one-column kernel
    shifted.copy_(aligned)
    assert shifted.data_ptr() % 16 != 0 and aligned.data_ptr() % 16 == 0
    for acc, mode in ((False, "time_evolution"), (True, "path_integral")):
        xf, xa = ops.diffuse(x0, aligned, 5, accumulate=acc)
        xs, xsa = ops.diffuse(x0, shifted, 5, accumulate=acc)
        got, got_s = ((xa, xsa) if acc else (xf, xs))
        np.testing.assert_allclose(got.cpu().numpy(), oracle.diffuse(x0, tr.astype(np.float64), 5, mode).ravel(), rtol=1e-11)
        assert torch.equal(got, got_s)
    # the graph-replayed long loop takes the same kernels
    xl, _ = ops.diffuse(x0, aligned, 40, accumulate=False)
    np.testing.assert_allclose(xl.cpu().numpy(), oracle.diffuse(x0, tr.astype(np.float64), 40, "time_evolution").ravel(), rtol=1e-10)


@pytest.mark.parametrize("edim", [1, 2, 3])
@pytest.mark.parametrize("compute", ["float64", "float32"])
def test_markov_factored_matches_dense_chain(ops, oracle, edim, compute):
    """prepare_markov / Diffusion.diffuse without the (n, n) matrix: the factors of vcy_prepare_markov_factored stepped by
    vcy_diffuse_step_factored (sparse product + Gauss transform on the fly) against the oracle's dense chain - both directions,
    a P that stores diagonal entries (replaced by the row maximum, analysis.py:1856), short loops and the graph-replayed long one,
    time evolution and path integral."""
    from scipy import sparse
    rng = np.random.default_rng(40 + edim)
    n, k = 700, 23
    emb = rng.normal(size=(n, edim)) * 3.0
    ix = np.stack([rng.choice(n, k, replace=False) for _ in range(n)])
    ix[::7, 0] = np.arange(n)[::7]                                    # some stored diagonal entries
    tp = np.zeros((n, n))
    np.put_along_axis(tp, ix, rng.random((n, k)) + 0.01, axis=1)
    tp /= tp.sum(1, keepdims=True)
    x0 = rng.random(n)
    cdt = torch.float64 if compute == "float64" else torch.float32
    rt = 1e-11 if compute == "float64" else 5e-5
    for direction in ("forward", "backwards"):
        ref = oracle.prepare_markov(tp, emb, 0.9, 1.6, direction)
        P = sparse.csr_matrix(tp if direction == "forward" else tp.T)
        P.sort_indices()
        fac = ops.prepare_markov_factored(P.indptr, P.indices, P.data, emb, 0.9, 1.6, compute_dtype=cdt)
        assert fac.shape == (n, n)
        np.testing.assert_allclose(fac.dense().cpu().numpy(), ref, rtol=1e-11, atol=1e-16)
        # the factors reassemble to the same matrix: tr = (0.2 K_W / kw + s) / tot
        d = np.sqrt(((emb[:, None, :] - emb[None, :, :]) ** 2).sum(-1))
        kw_full = np.exp(-d ** 2 / (2 * 1.6 ** 2)) / np.sqrt(2 * np.pi * 1.6 ** 2)
        s = sparse.csc_matrix((fac.scsc.cpu().numpy(), fac.rowidx.cpu().numpy(), fac.colptr.cpu().numpy()), shape=(n, n)).toarray()
        rebuilt = (0.2 * kw_full / fac.kw.cpu().numpy()[:, None] + s) / fac.tot.cpu().numpy()[:, None]
        np.testing.assert_allclose(rebuilt, ref, rtol=1e-11, atol=1e-16)
        for acc, mode in ((False, "time_evolution"), (True, "path_integral")):
            for steps in (3, 41):
                xf, xa = ops.diffuse(x0 / x0.sum(), fac, steps, accumulate=acc)
                want = oracle.diffuse(x0, ref, steps, mode).ravel()
                np.testing.assert_allclose((xa if acc else xf).cpu().numpy(), want, rtol=rt)


@pytest.mark.parametrize("edim", [1, 2, 3])
@pytest.mark.parametrize("compute", ["float64", "float32"])
def test_markov_culled_gauss_transform_equals_the_full_one(ops, oracle, edim, compute):
    """vcy_diffuse_step_factored_culled (cells sorted along the Hilbert curve, source runs out of the kernel's reach skipped)
    against vcy_diffuse_step_factored on clustered embeddings much wider than sigma_W - what prepare_markov is usually given - and
    against the oracle's dense chain on a small one; the dropped terms are below the accumulation's own rounding."""
    from scipy import sparse
    rng = np.random.default_rng(77 + edim)
    n, k = 6000, 12
    centres = rng.normal(size=(9, edim)) * 40.0
    lab = np.sort(rng.integers(0, 9, n))
    emb = centres[lab] + rng.normal(size=(n, edim)) * 2.0
    first, count = np.searchsorted(lab, lab), np.bincount(lab)[lab]      # transitions go to cells of the same cluster (a row whose
    ix = first[:, None] + (rng.integers(0, 1 << 30, (n, k)) % count[:, None])   # neighbours are all out of K_D's reach has no chain)
    ix[:, 0] = np.arange(n)                                              # (repeats allowed; the stored diagonal keeps every row alive)
    tp = rng.random((n, k)) + 0.01
    tp /= tp.sum(1, keepdims=True)
    indptr = np.arange(0, n * k + 1, k)
    cdt = torch.float64 if compute == "float64" else torch.float32
    full = ops.prepare_markov_factored(indptr, ix.ravel(), tp.ravel(), emb, 1.0, 0.5, compute_dtype=cdt, cull=False)
    cul = ops.prepare_markov_factored(indptr, ix.ravel(), tp.ravel(), emb, 1.0, 0.5, compute_dtype=cdt)
    assert full.cull is None and cul.cull is not None                     # auto: the embedding is ~200 kernel widths across
    x0 = rng.random(n)
    x0 /= x0.sum()
    for steps, acc in ((1, False), (7, True), (40, False)):
        a, aa = ops.diffuse(x0, full, steps, accumulate=acc)
        b, bb = ops.diffuse(x0, cul, steps, accumulate=acc)
        ra, rb = (aa, bb) if acc else (a, b)
        tol = 1e-12 if compute == "float64" else 2e-6
        assert float(((ra - rb).abs() / ra.abs().clamp_min(1e-300)).max()) < tol
        assert abs(float(rb.sum()) - (steps if acc else 1.0)) < (1e-9 if compute == "float64" else 1e-4)
    # a narrow compact embedding: nothing can be skipped, auto leaves the plain transform on
    near = ops.prepare_markov_factored(indptr, ix.ravel(), tp.ravel(), rng.normal(size=(n, edim)), 1.0, 2.0, compute_dtype=cdt)
    assert near.cull is None
    # and against the oracle's dense chain (forced culling on a small wide problem)
    m = 500
    embs = np.concatenate([rng.normal(size=(m // 2, edim)), rng.normal(size=(m - m // 2, edim)) + 30.0])
    tpd = np.zeros((m, m))
    np.put_along_axis(tpd, np.stack([rng.choice(m, 9, replace=False) for _ in range(m)]), rng.random((m, 9)) + 0.01, axis=1)
    tpd /= tpd.sum(1, keepdims=True)
    ref = oracle.prepare_markov(tpd, embs, 25.0, 0.7, "forward")          # (K_D wide: the random transitions cross between the blobs)
    P = sparse.csr_matrix(tpd)
    P.sort_indices()
    fac = ops.prepare_markov_factored(P.indptr, P.indices, P.data, embs, 25.0, 0.7, compute_dtype=cdt, cull=True)
    xs = rng.random(m)
    got, _ = ops.diffuse(xs / xs.sum(), fac, 9, accumulate=False)
    np.testing.assert_allclose(got.cpu().numpy(), oracle.diffuse(xs, ref, 9, "time_evolution").ravel(), rtol=1e-11 if compute == "float64" else 5e-5)


@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("transform,psc", [("sqrt", 1e-10), ("log10", 1e-10), ("linear", 0.0)])
@pytest.mark.parametrize("C,G,nr", [(200, 3100, 24), (40, 700, 9), (20, 500, 5)])
def test_coldeltacor_partial_dual_equals_two_launches(ops, dtype, transform, psc, C, G, nr):
    """vcy_coldeltacor_partial_dual (real + randomised control in one pass, analysis.py:1539-1542, 1578-1601) returns
    what two single launches return - on the grouped path (first two shapes: full and ragged chunks) and on the
    small-problem fallback (third)."""
    rng = np.random.default_rng(C * 7 + G)
    e = rng.gamma(2.0, 1.0, (G, C)) * (rng.random((G, C)) < 0.5)
    d = rng.normal(size=(G, C))
    d2 = np.sign(rng.normal(size=(G, C))) * rng.permuted(d, axis=1)          # the control: rows shuffled over cells, random sign
    d2[:, 3] = 0.0                                                           # a zero-variance control column -> NaN there only
    ixs = np.stack([rng.choice(C, nr, replace=False) for _ in range(C)])
    E, D, D2 = (ops.CellMatrix.from_genes_major(a, dtype) for a in (e, d, d2))
    tr = ops.TRANSFORMS[transform]
    a1 = ops.coldeltacor_partial(E, D, ixs, tr, ops.RULES_PARTIAL, psc)
    a2 = ops.coldeltacor_partial(E, D2, ixs, tr, ops.RULES_PARTIAL, psc)
    b1, b2 = ops.coldeltacor_partial_dual(E, D, D2, ixs, tr, ops.RULES_PARTIAL, psc)
    for x, y in ((a1, b1), (a2, b2)):
        x, y = x.cpu().numpy(), y.cpu().numpy()
        assert np.array_equal(np.isnan(x), np.isnan(y))
        # same arithmetic per element; the dual kernel sums a pair's moments in chunks of 1024 instead of 1536 genes
        np.testing.assert_allclose(x[~np.isnan(x)], y[~np.isnan(y)], rtol=0, atol=2e-6 if dtype == "float32" else 1e-13)
    self3 = torch.as_tensor(ixs[3] == 3, device=b1.device)       # a cell listed as its own neighbour has zero variance in A: NaN in both
    assert bool(torch.isnan(b2[3]).all()) and torch.equal(torch.isnan(b1[3]), self3)
    # a schedule over a subset leaves the other rows of both outputs untouched
    o1 = torch.full_like(b1, 7.0)
    o2 = torch.full_like(b2, 9.0)
    sub = torch.arange(0, C, 2, dtype=torch.int32)
    ops.coldeltacor_partial_dual(E, D, D2, ixs, tr, ops.RULES_PARTIAL, psc, order=sub, out=o1, out_rndm=o2)
    assert bool((o1[1::2] == 7.0).all()) and bool((o2[1::2] == 9.0).all())
    m = ~torch.isnan(b1[0::2])          # (half the cells may fall below the grouped kernel's minimum: another summation order)
    assert torch.allclose(o1[0::2][m], b1[0::2][m], atol=CORR_ATOL[dtype], rtol=0)


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_coldeltacor_partial_fused_dual(ops, dtype):
    """Velocity chain folded into the staging + randomised control, one launch == the three-kernel route."""
    rng = np.random.default_rng(77)
    C, G, nr = 96, 2000, 16
    S = rng.gamma(2.0, 1.0, (G, C)) * (rng.random((G, C)) < 0.6)
    U = rng.gamma(1.0, 1.0, (G, C)) * (rng.random((G, C)) < 0.6)
    d2 = rng.normal(size=(G, C))
    gam = torch.as_tensor(rng.gamma(2.0, 0.3, G), dtype=torch.float32)
    q = torch.as_tensor(rng.gamma(1.0, 0.05, G), dtype=torch.float32)
    ixs = np.stack([rng.choice(C, nr, replace=False) for _ in range(C)])
    Sx, Ux, D2 = (ops.CellMatrix.from_genes_major(a, dtype) for a in (S, U, d2))
    dmat = ops.velocity_chain(Sx, Ux, gam, q, want=("dmat",), transform=ops.SQRT, psc=1e-10)["dmat"]
    r1, r2 = ops.coldeltacor_partial_dual(Sx, dmat, D2, ixs, ops.SQRT, ops.RULES_PARTIAL, 1e-10)
    f1, f2 = ops.coldeltacor_partial_fused_dual(Sx, Ux, gam, q, D2, ixs, ops.SQRT, ops.RULES_PARTIAL, 1e-10)
    eq = lambda a, b: torch.equal(torch.nan_to_num(a, nan=7.0), torch.nan_to_num(b, nan=7.0))    # (self pairs are NaN in both)
    assert eq(r1, f1) and eq(r2, f2)                             # same kernel, d[c] staged from dmat or evaluated on the fly


def test_knn_search_segmented_equals_one_launch(ops, monkeypatch):
    """Point sets beyond one launch's candidate range are searched in segments and merged (ops._knn_search_segmented):
    same neighbours, same order (ties by index), same fp64 distances as the one-launch search."""
    rng = np.random.default_rng(3)
    C, P, k = 3000, 7, 12
    X = rng.normal(size=(C, P))
    X[100] = X[7]; X[2000] = X[7]; X[2999] = X[1500]          # exact duplicates across segments: distance ties at 0
    X[2996] = X[2992]; X[2991] = X[2992]                        # ... and inside a short last segment
    X[:, 3] = np.round(X[:, 3], 1)                               # and many exact ties at positive distances
    ref_i, ref_d = ops.knn_search(X, k, include_self=False)
    ref_is, ref_ds = ops.knn_search(X, k, include_self=True, q0=500, Q=700)
    for seg in (1024, 999, 2990):                                # the last one leaves a segment shorter than k + 1
        monkeypatch.setattr(ops, "KNN_SEGMENT", seg)
        i, d = ops.knn_search(X, k, include_self=False)
        assert torch.equal(i, ref_i) and torch.equal(d, ref_d), seg
        i, d = ops.knn_search(X, k, include_self=True, q0=500, Q=700)
        assert torch.equal(i, ref_is) and torch.equal(d, ref_ds), seg


def test_knn_search_pruned_is_exact(ops):
    """Projection-pruned search (atlas scale) == the brute-force search: indices, order (ties by index) and fp64 distances -
    on data whose leading coordinates separate the points (pruning bites), with duplicates and exact ties, and on isotropic
    data where the radius has to grow (the verification step), for a query range in the middle of the point set."""
    rng = np.random.default_rng(9)
    C, P, k = 20000, 12, 15
    X = np.concatenate([rng.uniform(0, 40, (C, 2)), rng.normal(0, 0.3, (C, P - 2))], 1)
    X[500] = X[40]; X[9000] = X[40]
    X[:, 5] = np.round(X[:, 5], 1)
    ref_i, ref_d = ops.knn_search(X, k, include_self=False)
    st = {}
    i, d = ops.knn_search_pruned(X, k, tile=1024, stats=st)
    assert torch.equal(i, ref_i) and torch.equal(d, ref_d)
    assert st["distance_evaluations"] < 0.25 * st["brute_force_evaluations"], st
    i, d = ops.knn_search_pruned(X, k, q0=3000, Q=5000, tile=700)
    assert torch.equal(i, ref_i[3000:8000]) and torch.equal(d, ref_d[3000:8000])
    seg = ops.PRUNED_SEGMENT                             # candidate sets beyond one launch: searched in pieces and merged
    try:
        ops.PRUNED_SEGMENT = 1500
        i, d = ops.knn_search_pruned(X, k, q0=100, Q=9000, tile=2048)
    finally:
        ops.PRUNED_SEGMENT = seg
    assert torch.equal(i, ref_i[100:9100]) and torch.equal(d, ref_d[100:9100])
    Y = rng.normal(size=(6000, 10))                      # no preferred directions: the projection bound is weak, results still exact
    ri, rd = ops.knn_search(Y, 8, include_self=False)
    st2 = {}
    i, d = ops.knn_search_pruned(Y, 8, tile=512, stats=st2)
    assert torch.equal(i, ri) and torch.equal(d, rd)


# ---------------------------------------------------------------- the no-pseudocount form of the partial sqrt rule (f32)
def _nopsc_problem(seed, G, C, nr, scale=1.0):
    rng = np.random.default_rng(seed)
    e = rng.gamma(2.0, 1.0, (G, C)) * (rng.random((G, C)) < 0.6) * scale
    e[:, 1] = e[:, 0]                                 # identical cells: zero-variance pair (NaN in every rule)
    e[: G // 2, 3] = e[: G // 2, 2]                   # half of the genes agree exactly: the zero rule matters
    d = rng.normal(size=(G, C))
    ixs = np.stack([rng.choice(C, nr, replace=False) for _ in range(C)])
    ixs[0, 0] = 1
    return e, d, ixs


@pytest.mark.parametrize("G,C,nr", [(3000, 64, 24), (1537, 40, 9), (50, 33, 8)])
def test_partial_nopsc_rule_against_the_literal_rule(ops, oracle, G, C, nr):
    """VCY_RULES_PARTIAL_NOPSC (f32, sqrt: A = sign(t) sqrt|t| by v_rsq_f32 + v_mul_legacy_f32) against the literal rule
    of speedboosted.pyx:372-378 with the default pseudocount: same NaN pattern, correlations within 2e-6 of the literal f32
    kernel and within the f32 tolerance of the f64 oracle; single and dual-control kernels, grouped and small-problem paths."""
    e, d, ixs = _nopsc_problem(G + C, G, C, nr)
    want = oracle.coldeltacor_partial_compact(e, d, ixs, "sqrt", 1e-10)
    E, D = ops.CellMatrix.from_genes_major(e, "float32"), ops.CellMatrix.from_genes_major(d, "float32")
    assert ops.partial_rules_for(E, ops.SQRT, 1e-10) == ops.RULES_PARTIAL_NOPSC
    lit = ops.coldeltacor_partial(E, D, ixs, ops.SQRT, ops.RULES_PARTIAL, 1e-10).cpu().numpy()
    fast = ops.coldeltacor_partial(E, D, ixs, ops.SQRT, ops.RULES_PARTIAL_NOPSC, 1e-10).cpu().numpy()
    assert np.array_equal(np.isnan(fast), np.isnan(want)) and np.array_equal(np.isnan(lit), np.isnan(want))
    assert np.isnan(fast[0, 0])
    ok = np.isfinite(want)
    assert np.abs(fast[ok] - lit[ok]).max() < 2e-6
    np.testing.assert_allclose(fast[ok], want[ok], atol=CORR_ATOL["float32"])
    D2 = ops.CellMatrix.from_genes_major(d[::-1].copy(), "float32")
    a, b = ops.coldeltacor_partial_dual(E, D, D2, ixs, ops.SQRT, ops.RULES_PARTIAL_NOPSC, 1e-10)
    b1 = ops.coldeltacor_partial(E, D2, ixs, ops.SQRT, ops.RULES_PARTIAL_NOPSC, 1e-10)
    assert torch.allclose(a.nan_to_num(7.0), torch.from_numpy(fast).to(a.device).nan_to_num(7.0), atol=2e-6)
    assert torch.allclose(b.nan_to_num(7.0), b1.nan_to_num(7.0), atol=2e-6)
    sub = torch.arange(0, C, 3, dtype=torch.int32)    # a schedule over a third of the cells: the small-problem kernel
    out = torch.full((C, nr), 5.0, dtype=torch.float32, device=E.t.device)
    ops.coldeltacor_partial(E, D, ixs, ops.SQRT, ops.RULES_PARTIAL_NOPSC, 1e-10, order=sub, out=out)
    got = out.cpu().numpy()[::3]
    sel = np.isfinite(want[::3])
    np.testing.assert_allclose(got[sel], want[::3][sel], atol=CORR_ATOL["float32"])


def test_partial_rules_for_keeps_the_literal_rule_when_the_pseudocount_counts(ops, oracle):
    """The helper that picks the rule: literal for f64, for log10 / linear, for a pseudocount above 1e-9 and for a matrix whose
    scale the pseudocount is NOT negligible against - where the two rules really differ and only the literal one matches
    the oracle."""
    e, d, ixs = _nopsc_problem(5, 2000, 48, 16, scale=1e-8)
    E32, D32 = ops.CellMatrix.from_genes_major(e, "float32"), ops.CellMatrix.from_genes_major(d, "float32")
    E64 = ops.CellMatrix.from_genes_major(e, "float64")
    big = ops.CellMatrix.from_genes_major(e * 1e8, "float32")
    assert ops.partial_rules_for(E32, ops.SQRT, 1e-10) == ops.RULES_PARTIAL            # tiny scale
    assert ops.partial_rules_for(big, ops.SQRT, 1e-10) == ops.RULES_PARTIAL_NOPSC
    assert ops.partial_rules_for(big, ops.SQRT, 1e-6) == ops.RULES_PARTIAL             # pseudocount not negligible
    speck = e * 1e8
    speck[5, 7] = 1e-25                                                                # one denormal-scale entry: v_rsq_f32 territory
    assert ops.partial_rules_for(ops.CellMatrix.from_genes_major(speck, "float32"), ops.SQRT, 1e-10) == ops.RULES_PARTIAL
    assert ops.partial_rules_for(big, ops.LOG10, 1e-10) == ops.RULES_PARTIAL
    assert ops.partial_rules_for(E64, ops.SQRT, 1e-10) == ops.RULES_PARTIAL
    want = oracle.coldeltacor_partial_compact(e, d, ixs, "sqrt", 1e-10)
    ok = np.isfinite(want)
    lit = ops.coldeltacor_partial(E32, D32, ixs, ops.SQRT, ops.partial_rules_for(E32, ops.SQRT, 1e-10), 1e-10).cpu().numpy()
    np.testing.assert_allclose(lit[ok], want[ok], atol=CORR_ATOL["float32"])
    fast = ops.coldeltacor_partial(E32, D32, ixs, ops.SQRT, ops.RULES_PARTIAL_NOPSC, 1e-10).cpu().numpy()
    # this is the case the helper guards: at |t| ~ 1e-8 the pseudocount is 1 % of a difference and dropping it shows
    assert np.abs(fast[ok] - want[ok]).max() > 4 * max(np.abs(lit[ok] - want[ok]).max(), 1e-6)


def test_partial_rules_decision_is_a_fact_about_the_whole_matrix(ops, monkeypatch):
    """ops.partial_rules_for reduces EVERY entry (vcy_abs_stats): one sub-1e-20 or denormal entry anywhere - in particular in a
    row a strided sample of 64 rows would never visit - forces the literal rule (v_rsq_f32 reads a denormal difference as zero:
    t * rsq|t| = inf would poison the pair's correlation).  Hypothesis places the speck."""
    from hypothesis import given, settings, strategies as hst
    G, C = 257, 1000                                  # stride of the old sample: C // 64 = 15 -> rows 0, 15, 30, ...
    rng = np.random.default_rng(11)
    base = (rng.gamma(2.0, 1.0, (C, G)) * (rng.random((C, G)) < 0.6)).astype(np.float32)
    E = ops.CellMatrix.from_cells_major(base, "float32")
    st = ops.abs_stats(E).cpu().numpy()
    nz = base != 0
    np.testing.assert_allclose(st[0], np.abs(base.astype(np.float64)).sum(), rtol=1e-12)
    assert st[1] == np.abs(base[nz]).min() and st[2] == nz.sum()
    assert ops.partial_rules_for(E, ops.SQRT, 1e-10) == ops.RULES_PARTIAL_NOPSC

    @settings(max_examples=25, deadline=None)
    @given(row=hst.integers(0, C - 1).filter(lambda r: r % 15 != 0), col=hst.integers(0, G - 1),
           val=hst.sampled_from([1e-21, 3e-30, 1e-38, 1.4e-45, -1e-25, -1.4e-45]))       # down to the smallest f32 denormal, either sign
    def planted(row, col, val):
        old = float(E.t[row, col])
        E.t[row, col] = val
        try:
            assert float(ops.abs_stats(E)[1]) == abs(np.float32(val))
            assert ops.partial_rules_for(E, ops.SQRT, 1e-10) == ops.RULES_PARTIAL
        finally:
            E.t[row, col] = old
    planted()
    # the padding columns never enter (they are zero) and an all-zero matrix reports +inf / 0
    Z = ops.CellMatrix.from_cells_major(np.zeros((5, 70), np.float32), "float32")
    z = ops.abs_stats(Z).cpu().numpy()
    assert z[0] == 0 and np.isinf(z[1]) and z[2] == 0 and ops.partial_rules_for(Z, ops.SQRT, 1e-10) == ops.RULES_PARTIAL
    # f64 matrices: same facts (the rule itself stays literal there)
    E64 = ops.CellMatrix.from_cells_major(base.astype(np.float64) * 1e-300, "float64")
    s64 = ops.abs_stats(E64).cpu().numpy()
    assert s64[2] == nz.sum() and s64[1] == (np.abs(base[nz].astype(np.float64)) * 1e-300).min()
    # explicit opt-outs
    assert ops.partial_rules_for(E, ops.SQRT, 1e-10, literal=True) == ops.RULES_PARTIAL
    monkeypatch.setenv("VELOCYTO_AMD_LITERAL_RULE", "1")
    assert ops.partial_rules_for(E, ops.SQRT, 1e-10) == ops.RULES_PARTIAL


@pytest.mark.parametrize("scale", [1e-3, 1e-4, 1e-5, 1e-6])
def test_partial_nopsc_rule_bound_on_scaled_matrices(ops, oracle, scale):
    """The a-priori bound stated in ops.partial_rules_for: |r_nopsc - r_literal| <= 2 ||delta||_2 / ||A - mean A||_2 with
    delta_g = sqrt(|t_g| + psc) - sqrt|t_g| (first order), on matrices scaled down towards the pseudocount; and the helper's
    decision keeps every correlation within the f32 tolerance of the fp64 oracle at every scale."""
    psc = 1e-10
    e, d, ixs = _nopsc_problem(77, 1600, 40, 12, scale=scale)
    E, D = ops.CellMatrix.from_genes_major(e, "float32"), ops.CellMatrix.from_genes_major(d, "float32")
    e32 = E.to_genes_major(np.float64)                                       # the values the kernels see
    lit = ops.coldeltacor_partial(E, D, ixs, ops.SQRT, ops.RULES_PARTIAL, psc).cpu().numpy().astype(np.float64)
    fast = ops.coldeltacor_partial(E, D, ixs, ops.SQRT, ops.RULES_PARTIAL_NOPSC, psc).cpu().numpy().astype(np.float64)
    want = oracle.coldeltacor_partial_compact(e32, D.to_genes_major(np.float64), ixs, "sqrt", psc)
    ok = np.isfinite(want)
    assert np.array_equal(np.isnan(fast), ~ok)
    worst_ratio = 0.0
    for c in range(0, e.shape[1], 3):
        for n in range(ixs.shape[1]):
            if not ok[c, n]:
                continue
            t = np.abs(e32[:, ixs[c, n]] - e32[:, c])
            A = np.sqrt(t)
            delta = np.where(t > 0, np.sqrt(t + psc) - A, 0.0)
            bound = 2.0 * np.linalg.norm(delta) / np.linalg.norm(A - A.mean())
            diff = abs(fast[c, n] - lit[c, n])
            assert diff <= 1.5 * bound + 4e-6, (scale, c, n, diff, bound)            # + the f32 rounding of two single-pass sums
            worst_ratio = max(worst_ratio, diff / max(bound, 1e-30))
    rules = ops.partial_rules_for(E, ops.SQRT, psc)
    assert rules == (ops.RULES_PARTIAL_NOPSC if np.abs(e32).mean() >= ops.SCALE_ORDINARY else ops.RULES_PARTIAL)
    chosen = fast if rules == ops.RULES_PARTIAL_NOPSC else lit
    np.testing.assert_allclose(chosen[ok], want[ok], atol=CORR_ATOL["float32"])


def test_partial_nopsc_is_rejected_where_it_is_not_defined(ops):
    e, d, ixs = _nopsc_problem(9, 64, 40, 8)
    E64, D64 = ops.CellMatrix.from_genes_major(e, "float64"), ops.CellMatrix.from_genes_major(d, "float64")
    E32, D32 = ops.CellMatrix.from_genes_major(e, "float32"), ops.CellMatrix.from_genes_major(d, "float32")
    with pytest.raises(ValueError, match="NOPSC"):
        ops.coldeltacor_partial(E64, D64, ixs, ops.SQRT, ops.RULES_PARTIAL_NOPSC, 1e-10)
    with pytest.raises(ValueError, match="NOPSC"):
        ops.coldeltacor_partial(E32, D32, ixs, ops.LOG10, ops.RULES_PARTIAL_NOPSC, 1e-10)


def test_f64_sqrt_element_accuracy_and_domain(ops, oracle):
    """The f64 partial-sqrt element seeds its square root from the f32 unit (sqrt_normal_f64: v_rsq_f32 + one Goldschmidt step +
    one residual correction, <= 2 ulp against a correctly rounded sqrt - not bit-equal to it, stated in DESIGN.md 3).  (1) Over a
    matrix whose differences sweep 30 decades, correlations stay within the f64 bar (1e-10) of the oracle's correctly rounded arithmetic;
    (2) the element is defined inside the f32 exponent range only: a matrix reaching 1e38 is refused by the callers' checks
    (ops.partial_rules_for, validate=True) instead of being evaluated with silent zeros."""
    rng = np.random.default_rng(3)
    G, C, nr = 900, 40, 9
    e = rng.gamma(2.0, 1.0, (G, C)) * 10.0 ** rng.integers(-15, 15, (G, 1))
    d = rng.normal(size=(G, C))
    ixs = np.stack([rng.choice(C, nr, replace=False) for _ in range(C)])
    E, Dm = ops.CellMatrix.from_genes_major(e, "float64"), ops.CellMatrix.from_genes_major(d, "float64")
    got = ops.coldeltacor_partial(E, Dm, ixs, ops.SQRT, ops.RULES_PARTIAL, 1e-10).cpu().numpy()
    ref = oracle.coldeltacor_partial_compact(e, d, ixs, "sqrt", 1e-10)
    _nan_close(got, ref, CORR_ATOL["float64"])
    assert ops.partial_rules_for(E, ops.SQRT, 1e-10) == ops.RULES_PARTIAL
    e[5, 7] = 2e38
    E = ops.CellMatrix.from_genes_major(e, "float64")
    with pytest.raises(ValueError, match="outside the supported range"):
        ops.coldeltacor_partial(E, Dm, ixs, ops.SQRT, ops.RULES_PARTIAL, 1e-10)
    with pytest.raises(ValueError, match="outside the supported range"):
        ops.partial_rules_for(E, ops.SQRT, 1e-10)
    ops.coldeltacor_partial(E, Dm, ixs, ops.LINEAR, ops.RULES_PARTIAL, 0.0)                  # other transforms have no such limit


def test_f64_exp2_element_accuracy(ops):
    """The fp64 Gauss-transform element of run_markov (csrc/misc.hip exp2_neg_tab: 2^(-d2) = 2^i T[j] 2^r from a 64-entry table and a
    degree-5 polynomial, 11 f64 instructions where the degree-13 form took 17; exp2_neg_tab256 in the culled kernel) against numpy's exp2, element by element, through the real
    step kernels: one source cell at the origin with weight 1, targets at distance sqrt(d2) - y[j] = coef 2^(-d2_j) and nothing else.
    Within 1 ulp of exp2 plus the one rounding of the product with coef over [0, 1000]; beyond the clamp the value is below 1e-300
    (the reference's exp() is 0 there: both vanish from any sum); both the plain and the culled transform."""
    dev = ops.require_gpu()
    L = ops._lib.lib()
    rng = np.random.default_rng(12)
    n = 1 << 15
    d2 = np.concatenate([[0.0, 1e-300, 2.0 ** -60, 1.0 / 64, 1.0, 1000.0, 1e-9], rng.uniform(0, 40, n // 2), 10.0 ** rng.uniform(-12, 3, n - n // 2 - 9), [1001.0, 5e4]])
    assert d2.size == n
    es = np.sqrt(d2)
    d2 = es * es                                            # the kernel's own argument: fl(df * df), df = es_j - 0
    es[0] = 0.0                                             # cell 0 is the source
    sigma_W = 0.7
    coef = 0.2 / np.sqrt(2.0 * np.pi * sigma_W * sigma_W)
    x = torch.zeros(n, dtype=torch.float64, device=dev)
    x[0] = 1.0
    ones = torch.ones(n, dtype=torch.float64, device=dev)
    colptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)               # no sparse part
    rowidx = torch.zeros(1, dtype=torch.int32, device=dev)
    scsc = torch.zeros(1, dtype=torch.float64, device=dev)
    es_t = torch.from_numpy(es.reshape(n, 1).copy()).to(dev)
    ws = torch.empty(int(L.vcy_markov_factored_workspace_bytes(n)), dtype=torch.uint8, device=dev)
    y = torch.zeros(n, dtype=torch.float64, device=dev)
    ops._lib.check(L.vcy_diffuse_step_factored(x.data_ptr(), y.data_ptr(), None, colptr.data_ptr(), rowidx.data_ptr(), scsc.data_ptr(), ones.data_ptr(), ones.data_ptr(),
                                               es_t.data_ptr(), 1, sigma_W, ws.data_ptr(), n, 0, ops.F64, ops._stream()), "diffuse_step_factored")
    got = y.cpu().numpy()
    want = coef * np.exp2(-d2)
    inside = d2 <= 1000.0
    ulp = np.spacing(want[inside])
    err = np.abs(got[inside] - want[inside]) / ulp
    assert err.max() <= 2.0, (err.max(), d2[inside][np.argmax(err)])        # 1 ulp of the element + the rounding of coef * e + numpy's own
    assert np.mean(err > 1.0) < 0.02
    assert np.all(got[~inside] < 1e-300) and np.all(got[~inside] >= 0)
    assert got[0] == coef                                                   # 2^0 exactly
    # the culled transform evaluates the same element from a 256-entry table and a degree-4 polynomial (exp2_neg_tab256): same bar, with a cut so
    # wide that no source run is skipped
    fac = ops.MarkovFactors(None, es_t, 1.0, sigma_W, colptr, rowidx, scsc, ones, ones, es_t, torch.float64).enable_culling(cut=1e12)
    gc, _ = ops.diffuse(x, fac, 1, accumulate=False)
    gc = gc.cpu().numpy()
    errc = np.abs(gc[inside] - want[inside]) / ulp
    assert errc.max() <= 2.0 and np.mean(errc > 1.0) < 0.02, (errc.max(), d2[inside][np.argmax(errc)])
    assert np.all(gc[~inside] < 1e-300) and np.all(gc[~inside] >= 0) and gc[0] == coef
    # a NaN coordinate: the cell's own normalisation is NaN in a real chain (kw sums the same distances) - with kw given as NaN here every
    # target the cell reaches is NaN, as in the dense chain
    kw = ones.clone()
    kw[0] = float("nan")
    y2 = torch.zeros(n, dtype=torch.float64, device=dev)
    ops._lib.check(L.vcy_diffuse_step_factored(x.data_ptr(), y2.data_ptr(), None, colptr.data_ptr(), rowidx.data_ptr(), scsc.data_ptr(), ones.data_ptr(), kw.data_ptr(),
                                               es_t.data_ptr(), 1, sigma_W, ws.data_ptr(), n, 0, ops.F64, ops._stream()), "diffuse_step_factored")
    assert bool(torch.isnan(y2).all())


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("C,G,n", [(9, 70, 3), (100, 1003, 17), (333, 4100, 64), (70, 515, 256)])
def test_embedding_scaling_against_the_two_step_route(ops, dtype, C, G, n):
    """vcy_embedding_scaling (calculate_embedding_shift's expression scaling, analysis.py:1714-1719, 1726-1731, in one launch with the
    (genes, cells) estimates kept in registers) against (1) numpy in fp64 on the stored values and (2) the two-step route it replaces
    (vcy_knn_pool_w2 + vcy_row_cosproj): ragged last group, repeated neighbours in a list, a cell that lists itself, a zero estimate
    (NaN, like the reference's 0 / 0), single and dual control, natural and permuted schedules (same numbers), run-to-run identical."""
    rng = np.random.default_rng(C + n)
    tdt = getattr(torch, dtype)
    hi = ops.CellMatrix.from_genes_major(rng.gamma(1.0, 2.0, (G, C)), dtype)
    dS = ops.CellMatrix.from_genes_major(rng.normal(size=(G, C)), dtype)
    dR = ops.CellMatrix.from_genes_major(rng.normal(size=(G, C)), dtype)
    ixs = np.stack([rng.choice(C, n, replace=n > C) for _ in range(C)]).astype(np.int32)
    ixs[0, :min(n, 2)] = 0                                    # a repeated neighbour, and the cell itself
    w = rng.normal(size=(C, n)) * 0.1
    w2 = rng.normal(size=(C, n)) * 0.1
    w[1] = 0.0                                                # estim of cell 1 is exactly zero -> 0 / 0
    dev = hi.t.device
    W, W2 = torch.as_tensor(w, device=dev).to(tdt), torch.as_tensor(w2, device=dev).to(tdt)
    cos, cos2 = ops.embedding_scaling(hi, dS, ixs, W, dR, W2)
    Hs, Ds, Rs = hi.to_genes_major().T, dS.to_genes_major().T, dR.to_genes_major().T             # stored values, (C, G) fp64
    for got, ww, dd, zero_row in ((cos, W.double().cpu().numpy(), Ds, True), (cos2, W2.double().cpu().numpy(), Rs, False)):
        est = np.einsum("ck,ckg->cg", ww, Hs[ixs])
        with np.errstate(invalid="ignore", divide="ignore"):
            ref = (dd * est).sum(1) / np.sqrt((est ** 2).sum(1))
        g = got.cpu().numpy()
        assert bool(np.isnan(ref[1])) == zero_row and np.array_equal(np.isnan(g), np.isnan(ref))
        ok = ~np.isnan(ref)
        np.testing.assert_allclose(g[ok], ref[ok], rtol=1e-10 if dtype == "float64" else 2e-4, atol=1e-12 if dtype == "float64" else 2e-5)
    # the route it replaces
    indptr = torch.arange(0, (C + 1) * n, n, dtype=torch.int64, device=dev)
    e1, e2 = ops.knn_pool_w2(hi, indptr, ixs.reshape(-1), W.reshape(-1), W2.reshape(-1))
    old1, old2 = ops.row_cosproj(dS, e1).cpu().numpy(), ops.row_cosproj(dR, e2).cpu().numpy()
    tol = dict(rtol=1e-10, atol=1e-12) if dtype == "float64" else dict(rtol=2e-4, atol=2e-5)
    ok = ~np.isnan(old1)
    np.testing.assert_allclose(cos.cpu().numpy()[ok], old1[ok], **tol)
    np.testing.assert_allclose(cos2.cpu().numpy(), old2, **tol)
    # single control, a permuted schedule, determinism
    (c_single,) = ops.embedding_scaling(hi, dS, ixs, W)
    assert torch.equal(torch.nan_to_num(c_single, nan=7.0), torch.nan_to_num(cos, nan=7.0))
    perm = torch.as_tensor(rng.permutation(C).astype(np.int32), device=dev)
    c_perm, c2_perm = ops.embedding_scaling(hi, dS, ixs, W, dR, W2, order=perm)
    np.testing.assert_allclose(torch.nan_to_num(c_perm, nan=7.0).cpu().numpy(), torch.nan_to_num(cos, nan=7.0).cpu().numpy(), **tol)
    again = ops.embedding_scaling(hi, dS, ixs, W, dR, W2, order=perm)
    assert torch.equal(torch.nan_to_num(again[0], nan=7.0), torch.nan_to_num(c_perm, nan=7.0)) and torch.equal(again[1], c2_perm)
    # lists wider than one workgroup sorts: the caller is told to take the two-step route
    if n == 256:
        wide = np.concatenate([ixs, ixs[:, :1]], 1)
        assert ops.embedding_scaling(hi, dS, wide, torch.zeros((C, n + 1), dtype=tdt, device=dev)) is None


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_hip_kernels_against_the_reference_kernels_where_built(ops, oracle, reference_kernels, dtype):
    """The HIP stage-D kernels against the REFERENCE'S OWN Cython kernels (oracle/_ref: velocyto/speedboosted.pyx built with its own
    flags in the build container, shipped to the GPU box as a binary and run in a subprocess) on fresh random inputs: all three
    transforms, partial (compact lists) and full (all pairs).  A tree that should hold the module and does not FAILS here
    (conftest.reference_kernels); only a tree that never had it skips."""
    rng = np.random.default_rng(23)
    G, C, nr = 1300, 96, 14
    e = rng.gamma(1.0, 2.0, (G, C)) * (rng.random((G, C)) < 0.8)
    d = rng.normal(size=(G, C))
    ixs = np.stack([rng.choice(C, nr, replace=False) for _ in range(C)])
    E, Dm = ops.CellMatrix.from_genes_major(e, dtype), ops.CellMatrix.from_genes_major(d, dtype)
    es, ds = E.to_genes_major(), Dm.to_genes_major()           # the stored (possibly f32-rounded) values, as fp64
    atol = CORR_ATOL[dtype]
    for transform, kern, psc in (("sqrt", ops.SQRT, 1e-10), ("log10", ops.LOG10, 1.0), ("linear", ops.LINEAR, 0.0)):
        ref, _ = oracle.reference_coldeltacor(es, ds, ixs, transform, psc, threads=4)
        got = ops.coldeltacor_partial(E, Dm, ixs, kern, ops.RULES_PARTIAL, psc).cpu().numpy()
        _nan_close(got, ref, atol)
        full_ref, _ = oracle.reference_coldeltacor(es, ds, None, transform, psc, threads=4)
        full = ops.coldeltacor_full(E, Dm, kern, psc).cpu().numpy()
        off = ~np.eye(C, dtype=bool)
        okf = np.isfinite(full_ref) & off
        np.testing.assert_allclose(full[okf], full_ref[okf], atol=atol)
