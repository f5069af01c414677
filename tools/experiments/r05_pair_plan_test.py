"""GPU parity tests of the pair plan (vcy_coldeltacor_pair_plan, vcy_cell_moments, vcy_coldeltacor_partial_paired): every mirrored pair
(c, i) / (i, c) of a set of neighbour lists evaluated once (speedboosted.pyx:366-378 evaluates both; A(i, c) = -A(c, i) for the odd
transforms).  Every call goes through the C ABI on cuda:0.

What is pinned: the plan against a NumPy restatement of its rule; the paired launch against the oracle (atol 1e-10 f64 / 5e-5 f32) and -
bit for bit - against the same kernel run with a plan that pairs nothing: a mirror's value does not depend on which side evaluated it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

CORR_ATOL = {"float64": 1e-10, "float32": 5e-5}


@pytest.fixture(scope="module")
def ops():
    import velocyto_amd  # noqa: F401
    from velocyto_amd import ops as _ops
    _ops.require_gpu()
    return _ops


def ring_lists(rng, C, nr, reach, dup=False, self_pairs=False):
    """Lists drawn from the cells within `reach` positions on a ring: most listed pairs have their mirror listed too."""
    ixs = np.empty((C, nr), dtype=np.int64)
    for c in range(C):
        cand = (c + np.concatenate([np.arange(-reach, 0), np.arange(1, reach + 1)])) % C
        ixs[c] = rng.choice(cand, nr, replace=False)
    if dup:                       # a few duplicated entries and repeated partners
        for c in rng.choice(C, C // 5, replace=False):
            ixs[c, -1] = ixs[c, 0]
    if self_pairs:
        for c in rng.choice(C, C // 7, replace=False):
            ixs[c, 1] = c
    return ixs


def numpy_plan(ixs, cell0=0, launch_of=None):
    """The rule of vcy_coldeltacor_pair_plan, restated."""
    C, nr = ixs.shape
    first = [dict() for _ in range(C)]
    for c in range(C):
        for n in range(nr):
            first[c].setdefault(int(ixs[c, n]), n)
    plan = np.full((C, nr), -1, dtype=np.int32)
    for c in range(C):
        for n in range(nr):
            i = int(ixs[c, n]); il = i - cell0
            if not (0 <= il < C) or il == c:
                continue
            if launch_of is not None and not (launch_of[c] >= 0 and launch_of[c] == launch_of[il]):
                continue
            if first[c][i] != n or (cell0 + c) not in first[il]:
                continue
            lo, hi = min(c, il), max(c, il)
            owner = lo if (lo + hi) & 1 else hi
            plan[c, n] = first[il][cell0 + c] if owner == c else -2
    return plan


@pytest.mark.parametrize("C,nr,reach,cell0", [(64, 9, 8, 0), (300, 24, 20, 0), (90, 40, 30, 17), (50, 300, 200, 0)])
def test_pair_plan_against_its_rule(ops, C, nr, reach, cell0):
    rng = np.random.default_rng(C + nr)
    if nr > 2 * reach or 2 * reach >= C:
        ixs = np.stack([rng.choice(C, nr, replace=nr > C) for _ in range(C)])
    else:
        ixs = ring_lists(rng, C, nr, reach, dup=True, self_pairs=True)
    plan = ops.pair_plan(ixs + cell0, cell0).cpu().numpy()
    ref = numpy_plan(ixs + cell0, cell0)
    np.testing.assert_array_equal(plan, ref)
    # every handed-over pair has exactly one evaluating mirror, pointing back at it
    for c, n in zip(*np.nonzero(plan >= 0)):
        i = ixs[c, n]
        assert ixs[i, plan[c, n]] == c and plan[i, plan[c, n]] == -2
    assert (plan >= 0).sum() == (plan == -2).sum()
    if nr <= 2 * reach and 2 * reach < C:
        assert (plan != -1).mean() > 0.3                      # the case exercises what it is meant to
    # launch ids: cells pair within one launch only; negative ids never
    lo = rng.integers(-1, 2, C).astype(np.int32)
    plan2 = ops.pair_plan(ixs + cell0, cell0, torch.from_numpy(lo)).cpu().numpy()
    np.testing.assert_array_equal(plan2, numpy_plan(ixs + cell0, cell0, lo))


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("transform,psc", [("sqrt", 1e-10), ("linear", 0.0)])
@pytest.mark.parametrize("C,G,nr,reach", [(200, 3100, 24, 20), (64, 769, 9, 6), (120, 1536, 16, 10)])
def test_paired_launch_against_the_oracle_and_the_unpaired_kernel(ops, oracle, dtype, transform, psc, C, G, nr, reach):
    rng = np.random.default_rng(C * 3 + G)
    e = rng.gamma(2.0, 1.0, (G, C)) * (rng.random((G, C)) < 0.5)
    e[:, 5] = e[:, 4]                                             # identical cells: t == 0 everywhere, zero variance -> NaN both ways
    d = rng.normal(size=(G, C))
    d[:, 9] = 0.0                                                 # a zero-variance d column: NaN in its own row AND in its mirrors' slots
    ixs = ring_lists(rng, C, nr, reach, dup=True, self_pairs=True)
    E, D = ops.CellMatrix.from_genes_major(e, dtype), ops.CellMatrix.from_genes_major(d, dtype)
    tr = ops.TRANSFORMS[transform]
    got = ops.coldeltacor_partial_paired(E, D, ixs, tr, ops.RULES_PARTIAL, psc)
    ref = oracle.coldeltacor_partial_compact(e, d, ixs, transform, psc)
    g = got.cpu().numpy()
    ok = ~np.isnan(ref)
    assert np.array_equal(np.isnan(g), ~ok)
    np.testing.assert_allclose(g[ok], ref[ok], atol=CORR_ATOL[dtype], rtol=0)
    # the same kernel with a plan that pairs nothing: identical bits
    nothing = torch.full((C, nr), -1, dtype=torch.int32, device=got.device)
    alone = ops.coldeltacor_partial_paired(E, D, ixs, tr, ops.RULES_PARTIAL, psc, plan=nothing)
    assert torch.equal(torch.isnan(alone), torch.isnan(got))
    assert torch.equal(alone[~torch.isnan(alone)], got[~torch.isnan(got)])
    # and the plain entry point (other chunk length, d-moments summed while staging): same numbers to rounding
    plain = ops.coldeltacor_partial(E, D, ixs, tr, ops.RULES_PARTIAL, psc).cpu().numpy()
    np.testing.assert_allclose(g[ok], plain[ok], atol=1e-12 if dtype == "float64" else 2e-5, rtol=0)
    # scheduling order does not change a bit
    order = torch.from_numpy(rng.permutation(C).astype(np.int32))
    again = ops.coldeltacor_partial_paired(E, D, ixs, tr, ops.RULES_PARTIAL, psc, order=order)
    assert torch.equal(again[~torch.isnan(again)], got[~torch.isnan(got)])


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_paired_launch_subsets_and_launch_ids(ops, dtype):
    rng = np.random.default_rng(11)
    C, G, nr = 160, 2000, 20
    e = rng.gamma(2.0, 1.0, (G, C)) * (rng.random((G, C)) < 0.5)
    d = rng.normal(size=(G, C))
    ixs = ring_lists(rng, C, nr, 14)
    E, D = ops.CellMatrix.from_genes_major(e, dtype), ops.CellMatrix.from_genes_major(d, dtype)
    full = ops.coldeltacor_partial_paired(E, D, ixs, ops.SQRT, ops.RULES_PARTIAL, 1e-10)
    # a schedule over a subset: only those rows are written, and they equal the full launch's
    sub = torch.from_numpy(np.sort(rng.choice(C, 100, replace=False)).astype(np.int32))
    o = torch.full_like(full, 7.0)
    ops.coldeltacor_partial_paired(E, D, ixs, ops.SQRT, ops.RULES_PARTIAL, 1e-10, order=sub, out=o)
    rest = np.setdiff1d(np.arange(C), sub.numpy())
    assert bool((o[rest] == 7.0).all())
    assert torch.equal(o[sub.long()], full[sub.long()])
    # two launches that share one plan built with launch ids (the overlap schedule of a sharded run): together the full result
    ids = torch.from_numpy((np.arange(C) % 3 == 0).astype(np.int32))
    plan = ops.pair_plan(ixs, 0, ids)
    dm = ops.cell_moments(D)
    o2 = torch.full_like(full, 7.0)
    for k in (0, 1):
        order = torch.nonzero(ids == k).flatten().to(torch.int32)
        ops.coldeltacor_partial_paired(E, D, ixs, ops.SQRT, ops.RULES_PARTIAL, 1e-10, order=order, out=o2, plan=plan, dm=dm)
    assert torch.equal(o2, full)
    # a block of cells with its own rows of d (cell0 / d_row0 of a sharded rank): neighbours outside the block stay ordinary pairs
    blk = ops.coldeltacor_partial_paired(E, ops.CellMatrix(D.t[40:120].contiguous(), G), ixs[40:120], ops.SQRT, ops.RULES_PARTIAL, 1e-10, cell0=40, d_row0=40)
    assert torch.equal(blk, full[40:120])


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_paired_launch_on_wide_lists(ops, oracle, dtype):
    """Lists wider than one tile (column tiles, rows sorted by neighbour for the launch): a mirror may sit in another tile."""
    rng = np.random.default_rng(5)
    C, G, nr = 700, 300, 600
    e = rng.gamma(2.0, 1.0, (G, C)) * (rng.random((G, C)) < 0.5)
    d = rng.normal(size=(G, C))
    ixs = np.stack([rng.choice(C, nr, replace=False) for _ in range(C)])
    E, D = ops.CellMatrix.from_genes_major(e, dtype), ops.CellMatrix.from_genes_major(d, dtype)
    got = ops.coldeltacor_partial_paired(E, D, ixs, ops.SQRT, ops.RULES_PARTIAL, 1e-10).cpu().numpy()
    ref = oracle.coldeltacor_partial_compact(e, d, ixs, "sqrt", 1e-10)
    ok = ~np.isnan(ref)
    assert np.array_equal(np.isnan(got), ~ok)
    np.testing.assert_allclose(got[ok], ref[ok], atol=CORR_ATOL[dtype], rtol=0)


def test_paired_entry_falls_back_where_the_transform_is_not_odd(ops):
    rng = np.random.default_rng(3)
    C, G, nr = 64, 500, 12
    e = rng.gamma(2.0, 1.0, (G, C))
    d = rng.normal(size=(G, C))
    ixs = ring_lists(rng, C, nr, 8)
    E, D = ops.CellMatrix.from_genes_major(e, "float64"), ops.CellMatrix.from_genes_major(d, "float64")
    for tr, rules in ((ops.LOG10, ops.RULES_PARTIAL), (ops.SQRT, ops.RULES_FULL)):
        a = ops.coldeltacor_partial_paired(E, D, ixs, tr, rules, 1e-3)
        b = ops.coldeltacor_partial(E, D, ixs, tr, rules, 1e-3)
        assert torch.equal(torch.nan_to_num(a, nan=7.0), torch.nan_to_num(b, nan=7.0))
    # a problem below the grouped kernel's minimum (20 cells): the library reports it, the wrapper runs the plain kernel
    a = ops.coldeltacor_partial_paired(E, ops.CellMatrix(D.t[:20].contiguous(), G), ixs[:20] % 20, ops.SQRT, ops.RULES_PARTIAL, 1e-10)
    b = ops.coldeltacor_partial(E, ops.CellMatrix(D.t[:20].contiguous(), G), ixs[:20] % 20, ops.SQRT, ops.RULES_PARTIAL, 1e-10)
    assert torch.equal(torch.nan_to_num(a, nan=7.0), torch.nan_to_num(b, nan=7.0))
