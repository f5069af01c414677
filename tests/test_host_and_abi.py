"""CPU-side checks of the product: the C-ABI library loads and exports every symbol that
include/velocyto_hip.h declares, argument validation works without a GPU, and the host-side
balanced-kNN loop (C++) is bit-exact against the reference's golden vectors."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import velocyto_amd
    velocyto_amd.build()
    from velocyto_amd import _lib
    return _lib


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "velocyto_hip.h")).read()
    declared = set(re.findall(r"\b(vcy_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    L = lib.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in velocyto_hip.h but not exported"
    assert declared == set(lib.SIGNATURES), declared ^ set(lib.SIGNATURES)
    assert L.vcy_abi_version() == 4
    # and the maintainer's guide names every one of them (which reference call it replaces, or what a binder needs it for)
    guide = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert not [n for n in sorted(declared) if n not in guide]


def test_no_cpu_fallback_in_product():
    """The product never imports the oracle and has no numpy compute fallback."""
    pkg = os.path.join(ROOT, "velocyto.py_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn


def test_argument_validation_without_gpu(lib):
    L = lib.lib()
    rc = L.vcy_coldeltacor_partial(None, None, None, None, None, 4, 4, 4, 0, 4, 0, 2, 1, 1, 0.0, 0, None)
    assert rc == -1 and b"null pointer" in L.vcy_last_error()
    rc = L.vcy_balance_knn_host(None, None, None, None, 1, 1, 1, 1, 1, None, None, None)
    assert rc == -1


@pytest.mark.parametrize("tag", ["bal", "balc"])
def test_balance_knn_host_golden(lib, golden, tag):
    from velocyto_amd import ops
    g = golden("neighbors")
    dsi, dist = g[f"{tag}_dsi"], g[f"{tag}_dist"]
    groups = g["groups"] if tag == "balc" else None
    l0 = np.bincount(dsi.ravel(), minlength=dsi.shape[0])
    lsi = np.argsort(l0, kind="mergesort")[::-1]
    dist_new, dsi_new, l = ops.balance_knn_host(dsi, dist, lsi, groups, maxl=14, k=9)
    assert np.array_equal(dsi_new, g[f"{tag}_dsi_new"]) and np.array_equal(l, g[f"{tag}_l"])
    assert np.array_equal(dist_new, g[f"{tag}_dist_new"])


def test_balance_knn_host_padding(lib, golden):
    from velocyto_amd import ops
    g = golden("neighbors")
    dsi, dist = g["bal_dsi"][:, :12], g["bal_dist"][:, :12]
    lsi = np.argsort(np.bincount(dsi.ravel(), minlength=dsi.shape[0]), kind="mergesort")[::-1]
    d, i, l = ops.balance_knn_host(dsi, dist, lsi, None, maxl=4, k=9)
    assert np.array_equal(i, g["pad_dsi_new"]) and np.array_equal(l, g["pad_l"]) and np.array_equal(d, g["pad_dist_new"])
    dsi = g["bal_dsi"]
    lsi = np.argsort(np.bincount(dsi.ravel(), minlength=dsi.shape[0]), kind="mergesort")[::-1]
    d, i, l = ops.balance_knn_host(dsi, None, lsi, None, maxl=14, k=9)
    assert np.array_equal(i, g["nd_dsi_new"]) and np.array_equal(l, g["nd_l"]) and np.array_equal(d, g["nd_dist_new"])
    with pytest.raises(AssertionError):
        ops.balance_knn_host(dsi[:, :5], None, lsi, None, maxl=14, k=9)


def test_missing_library_fails_loudly(lib, monkeypatch, tmp_path):
    """No silent fallback: with the .so absent every entry point raises."""
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lib.lib()


def test_gpu_required_message():
    import torch
    from velocyto_amd import ops
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.require_gpu()
    with pytest.raises(RuntimeError):
        ops.CellMatrix.from_genes_major(np.zeros((3, 4)))


def test_balance_knn_properties_hypothesis(lib):
    """Invariants of the greedy balancing (neighbors.py:11-72) on random sight graphs: in-degree cap, no
    self neighbours besides column 0 / padding, neighbours taken in sight order, l consistent with dsi_new."""
    from hypothesis import given, settings, strategies as st
    from velocyto_amd import ops

    @settings(max_examples=40, deadline=None)
    @given(st.integers(5, 40), st.integers(1, 6), st.integers(1, 8), st.integers(0, 2 ** 31 - 1))
    def check(n, k, maxl, seed):
        rng = np.random.default_rng(seed)
        k = min(k, n - 2)
        K = int(rng.integers(k + 1, n + 1))                          # sight (incl. self) between k+1 and n
        dsi = np.stack([np.concatenate([[i], rng.permutation(np.delete(np.arange(n), i))[:K - 1]]) for i in range(n)])
        dist = np.sort(rng.random((n, K)), 1)
        dist[:, 0] = 0
        lsi = np.argsort(np.bincount(dsi.ravel(), minlength=n), kind="mergesort")[::-1]
        d, i, l = ops.balance_knn_host(dsi, dist, lsi, None, maxl, k)
        assert i.shape == (n, k + 1) and np.all(i[:, 0] == np.arange(n))
        real = i[:, 1:] != np.arange(n)[:, None]
        assert l.max(initial=0) <= maxl and l.sum() == real.sum()
        assert np.array_equal(np.bincount(i[:, 1:][real], minlength=n), l)
        for r in range(n):                                           # neighbours keep their sight order
            pos = [int(np.where(dsi[r] == m)[0][0]) for m in i[r, 1:][real[r]]]
            assert pos == sorted(pos)
    check()


def test_package_root_mirrors_the_reference_exports():
    """velocyto/__init__.py:12-15 re-exports these analysis-side names at the package root; so does velocyto_amd (lazily)."""
    import velocyto_amd as v
    for name in ("BalancedKNN", "convolve_by_sparse_weights", "fit_slope", "_fit1_slope", "clusters_stats", "dump_hdf5", "load_hdf5",
                 "VelocytoLoom", "ixs_thatsort_a2b", "load_velocyto_hdf5"):
        assert callable(getattr(v, name)), name
    for mod in ("estimation", "neighbors", "diffusion", "analysis", "speedboosted", "serialization"):
        assert getattr(v, mod).__name__ == f"velocyto_amd.{mod}"
    sb = v.speedboosted
    assert all(hasattr(sb, n) for n in ("_colDeltaCor", "_colDeltaCorSqrt", "_colDeltaCorLog10", "_colDeltaCorpartial",
                                         "_colDeltaCorSqrtpartial", "_colDeltaCorLog10partial"))


@pytest.mark.parametrize("n,k,diag", [(50, 7, 1), (200, 30, 1), (200, 30, 0.3), (64, 1, 2.5), (40, 39, 1), (7, 6, 1e-3)])
def test_weights_written_directly_equal_the_scipy_chain(n, k, diag):
    """knn_imputation's weights from a graph without zero distances (analysis.py:1006-1010) written out directly
    (neighbors.weights_from_sorted_knn) against the reference's scipy chain: same structure, same values to the bit."""
    import warnings
    from scipy import sparse
    from velocyto_amd.neighbors import weights_from_sorted_knn, connectivity_to_weights
    rng = np.random.default_rng(n + k)
    idx = np.stack([np.sort(rng.choice(np.delete(np.arange(n), c), k, replace=False)) for c in range(n)])
    dist = rng.random((n, k)) + 0.1
    perm = np.stack([rng.permutation(k) for _ in range(n)])                          # the reference's graph is nearest-first, not column-sorted
    knn = sparse.csr_matrix((np.take_along_axis(dist, perm, 1).ravel(), np.take_along_axis(idx, perm, 1).ravel(), np.arange(0, n * k + 1, k)), shape=(n, n))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        conn = (knn > 0).astype(float)
        conn.setdiag(diag)
    ref = connectivity_to_weights(conn)
    ref.sort_indices()
    got = weights_from_sorted_knn(idx.astype(np.int32), diag)
    assert np.array_equal(ref.indptr, got.indptr) and np.array_equal(ref.indices, got.indices) and np.array_equal(ref.data, got.data)


def _rng_state_equal(a, b):
    return a[0] == b[0] and np.array_equal(a[1], b[1]) and a[2:] == b[2:]


@pytest.mark.parametrize("n,size,cells,kind", [(101, 50, 40, "ramp"), (501, 250, 64, "ramp"), (30, 30, 25, "ramp"), (64, 20, 30, "zeros"),
                                               (17, 1, 50, "uniform"), (200, 199, 12, "steep"), (12, 0, 5, "uniform"), (9, 4, 0, "uniform"),
                                               (501, 250, 300, "ramp"), (4000, 40, 30, "plateau"), (4000, 900, 6, "plateau"), (2500, 30, 40, "uniform")])
def test_choice_stream_host_replays_numpy(lib, n, size, cells, kind):
    """The neighbour sampling of estimate_transition_prob (analysis.py:1561-1564): the block helper must return what the
    reference's per-cell np.random.choice(n, size, replace=False, p=p) calls return, draw for draw, and leave numpy's global
    RNG in the same state - whatever the block / pool sizes (refill path, cells that do not fit the pool, the short measuring
    block followed by full ones) and whichever lookup the library picks: bucket table + fixed scan in the first round (or a
    binary search when a bucket of the table is wide: "plateau", long runs of zero probability), normalised cdf + binary search
    in big later rounds, unnormalised sums + exact predicate in small ones."""
    from velocyto_amd import ops
    p = {"ramp": np.linspace(0.5, 0.1, n), "uniform": np.ones(n), "steep": np.geomspace(1.0, 1e-6, n),
         "zeros": np.where(np.arange(n) % 3 == 0, 0.0, np.linspace(1, 2, n)),
         "plateau": np.where((np.arange(n) > 100) & (np.arange(n) < 2900), 0.0, np.linspace(1, 2, n))}[kind]
    p = p / p.sum()
    np.random.seed(15071990 + n)
    want = np.stack([np.random.choice(n, size=(size,), replace=False, p=p) for _ in range(cells)], 0) if cells else np.empty((0, size), dtype=np.int64)
    after = np.random.get_state()
    tail = np.random.random_sample(3)
    for block, factor in ((4096, 1.5), (7, 1.5), (3, 0.2), (1, 0.01)):
        np.random.seed(15071990 + n)
        got = ops.choice_stream_host(n, size, p, cells, block=block, pool_factor=factor)
        assert got.dtype == np.int64 and got.shape == (cells, size)
        assert np.array_equal(got, want)
        assert _rng_state_equal(np.random.get_state(), after)
        assert np.array_equal(np.random.random_sample(3), tail)


@pytest.mark.parametrize("n,size,cells,kind,knobs", [
    (60, 20, 6000, "ramp", {}),                                                  # production settings: 8 workers, chains meet within the slack
    (60, 20, 3000, "steep", {"VCY_CHOICE_THREADS": "5"}),
    (40, 1, 4000, "uniform", {}),                                                # one uniform per cell: every chain is the true one
    (501, 250, 1400, "ramp", {"VCY_CHOICE_THREADS": "2"}),
    (30, 12, 700, "ramp", {"VCY_CHOICE_MIN_SHARE": "16", "VCY_CHOICE_PREFIX": "8"}),          # many tiny shares
    (30, 12, 700, "zeros", {"VCY_CHOICE_MIN_SHARE": "16", "VCY_CHOICE_PREFIX": "8", "VCY_CHOICE_SLACK": "0"}),   # no slack: chains are
    (90, 40, 900, "steep", {"VCY_CHOICE_MIN_SHARE": "40", "VCY_CHOICE_PREFIX": "3", "VCY_CHOICE_SLACK": "1"}),   # continued cell by cell until they meet
    (30, 12, 300, "ramp", {"VCY_CHOICE_MIN_SHARE": "1", "VCY_CHOICE_PREFIX": "1", "VCY_CHOICE_SLACK": "0", "VCY_CHOICE_THREADS": "64"}),
])
def test_choice_stream_parallel_chains_are_the_sequential_stream(lib, monkeypatch, n, size, cells, kind, knobs):
    """vcy_choice_stream_host splits the cells among threads that replay from GUESSED pool positions and stitches the true chain
    through the positions the chains share.  Against numpy's own per-cell calls on the same RandomState: every row, the number of
    uniforms consumed, and the same again when the pool ends early (fewer cells done, never a wrong one)."""
    import ctypes
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    p = {"ramp": np.linspace(0.5, 0.1, n), "uniform": np.ones(n), "steep": np.geomspace(1.0, 1e-4, n),
         "zeros": np.where(np.arange(n) % 3 == 0, 0.0, np.linspace(1, 2, n))}[kind]
    p = p / p.sum()
    rs = np.random.RandomState(77 + n)
    want = np.stack([rs.choice(n, size=size, replace=False, p=p) for _ in range(cells)])
    starts = None
    probe = np.random.RandomState(77 + n)                                           # how many uniforms those calls took
    big = probe.random_sample(cells * size * 8 + 64)
    L = lib.lib()
    out = np.full((cells, size), -1, dtype=np.int64)
    cd, used = ctypes.c_int64(0), ctypes.c_int64(0)
    assert L.vcy_choice_stream_host(big.ctypes.data, big.size, p.ctypes.data, n, size, cells, out.ctypes.data, ctypes.byref(cd), ctypes.byref(used)) == 0
    assert cd.value == cells and np.array_equal(out, want)
    check = np.random.RandomState(77 + n)
    check.random_sample(used.value)
    assert np.array_equal(check.random_sample(4), rs.random_sample(4))             # consumed exactly what the per-cell calls consumed
    total = used.value
    for frac in (0.9, 0.51, 0.05):                                                 # the pool ends inside the stream
        out2 = np.full((cells, size), -1, dtype=np.int64)
        cut = int(total * frac)
        assert L.vcy_choice_stream_host(big.ctypes.data, cut, p.ctypes.data, n, size, cells, out2.ctypes.data, ctypes.byref(cd), ctypes.byref(used)) == 0
        assert 0 <= cd.value < cells and used.value <= cut
        assert np.array_equal(out2[: cd.value], want[: cd.value])
        # the next cell really did not fit: replaying it alone from where the stream stopped runs out of uniforms
        one = np.full((1, size), -1, dtype=np.int64)
        cd1, u1 = ctypes.c_int64(0), ctypes.c_int64(0)
        L.vcy_choice_stream_host(big[used.value:].ctypes.data, cut - used.value, p.ctypes.data, n, size, 1, one.ctypes.data, ctypes.byref(cd1), ctypes.byref(u1))
        assert cd1.value == 0


def test_choice_stream_host_rewinds_the_rng_when_a_block_fails(lib):
    """An exception out of on_block (a stage-D launch of estimate_transition_prob failing) must not leave numpy's global RNG
    advanced by the uniforms prefetched for later blocks: afterwards it stands exactly where the per-cell np.random.choice calls
    for the cells finished so far would have left it."""
    from velocyto_amd import ops
    n, size, cells = 101, 50, 40
    p = np.linspace(0.5, 0.1, n)
    p = p / p.sum()
    for fail_at in (0, 2):
        seen = []

        def on_block(rows, c0, c1):
            seen.append((c0, c1))
            if len(seen) == fail_at + 1:
                raise RuntimeError("launch failed")
        np.random.seed(7)
        with pytest.raises(RuntimeError, match="launch failed"):
            ops.choice_stream_host(n, size, p, cells, block=7, on_block=on_block)
        done = seen[-1][1]
        state = np.random.get_state()
        np.random.seed(7)
        for _ in range(done):
            np.random.choice(n, size=(size,), replace=False, p=p)
        assert 0 < done < cells and _rng_state_equal(state, np.random.get_state())


def test_choice_stream_host_argument_errors(lib):
    from velocyto_amd import ops
    p = np.ones(10) / 10
    with pytest.raises(ValueError):
        ops.choice_stream_host(10, 11, p, 3)                       # larger sample than population
    with pytest.raises(ValueError):
        ops.choice_stream_host(10, 3, p * 0.9, 3)                  # does not sum to 1
    with pytest.raises(ValueError):
        ops.choice_stream_host(10, 3, p[:9], 3)
    q = np.zeros(10); q[:2] = 0.5
    with pytest.raises(ValueError):
        ops.choice_stream_host(10, 3, q, 3)                        # fewer non-zero entries in p than size
    state = np.random.get_state()
    q = p.copy(); q[0], q[1] = -0.1, 0.3
    with pytest.raises(ValueError):
        ops.choice_stream_host(10, 3, q, 3)                        # negative probability


def test_module_level_helpers_of_analysis():
    """analysis.py:2392-2420 under the reference's names (host functions): the sparse-array loop of scale_to_match_median and
    the numba-seeded row permutation, which without numba is the same sequence of numpy legacy draws."""
    import scipy.sparse as sp
    from velocyto_amd import analysis
    rng = np.random.default_rng(3)
    m = sp.random(30, 30, density=0.3, random_state=4, format="csr")
    tot = rng.random(30) + 0.1
    want = analysis.scale_to_match_median(m, tot)
    np.testing.assert_array_equal(analysis._scale_to_match_median(m.data, m.indices, m.indptr, tot), want.data)
    A = rng.normal(size=(7, 40))
    B = A.copy()
    analysis.numba_random_seed(11)
    analysis.permute_rows_nsign(A)
    np.random.seed(11)
    plmi = np.array([+1, -1])
    for i in range(B.shape[0]):
        np.random.shuffle(B[i, :])
        B[i, :] = B[i, :] * np.random.choice(plmi, size=B.shape[1])
    np.testing.assert_array_equal(A, B)
    np.testing.assert_allclose(np.sort(np.abs(A), 1), np.sort(np.abs(B), 1))


def test_serialization_object_codec():
    """serialization.py:9-41: non-array attributes travel as pickled + zlib-compressed uint8 datasets (same bytes as the reference
    writes: protocol 2, level 9)."""
    import pickle
    import zlib
    from velocyto_amd import serialization
    obj = {"a": [1, 2, 3], "b": ("x", 2.5), "c": np.arange(5)}
    u = serialization._obj2uint(obj)
    assert u.dtype == np.uint8 and u.ndim == 1
    assert u.tobytes() == zlib.compress(pickle.dumps(obj, protocol=2), 9)
    back = serialization._uint2obj(u)
    assert back["a"] == obj["a"] and back["b"] == obj["b"] and np.array_equal(back["c"], obj["c"])


def test_choice_stream_host_hypothesis(lib):
    """Random shapes, probability vectors (with zeros, heavy tails, near-ties) and block sizes: always numpy's own draws."""
    from hypothesis import given, settings, strategies as st
    from velocyto_amd import ops

    @settings(max_examples=60, deadline=None)
    @given(st.integers(1, 70), st.integers(0, 9), st.integers(0, 2 ** 31 - 1), st.sampled_from(["flat", "ramp", "spiky", "zeros"]),
           st.integers(1, 9), st.floats(0.01, 2.0))
    def run(n, cells, seed, kind, block, factor):
        rng = np.random.default_rng(seed)
        p = {"flat": np.ones(n), "ramp": np.linspace(1.0, 0.05, n), "spiky": rng.random(n) ** 12 + 1e-12,
             "zeros": np.where(rng.random(n) < 0.4, 0.0, rng.random(n) + 1e-3)}[kind]
        if not (p > 0).any():
            p[0] = 1.0
        p = p / p.sum()
        size = int(rng.integers(0, int((p > 0).sum()) + 1))
        np.random.seed(seed % (2 ** 32))
        want = np.stack([np.random.choice(n, size=(size,), replace=False, p=p) for _ in range(cells)], 0) if cells else np.empty((0, size), dtype=np.int64)
        after = np.random.get_state()
        np.random.seed(seed % (2 ** 32))
        got = ops.choice_stream_host(n, size, p, cells, block=block, pool_factor=factor)
        assert np.array_equal(got, want)
        assert _rng_state_equal(np.random.get_state(), after)

    run()


def test_choice_stream_parallel_chains_hypothesis(lib, monkeypatch):
    """Random populations, sample sizes, thread counts, shares, prefixes and slacks: the stitched parallel replay always returns
    numpy's own rows, consumes numpy's own number of uniforms, and stops at the same cell when the pool ends early."""
    import ctypes
    from hypothesis import given, settings, strategies as st
    L = lib.lib()

    @settings(max_examples=80, deadline=None)
    @given(st.integers(2, 40), st.integers(30, 260), st.integers(0, 2 ** 31 - 1), st.sampled_from(["flat", "ramp", "spiky", "zeros"]),
           st.integers(2, 24), st.integers(1, 24), st.integers(1, 12), st.integers(0, 30), st.floats(0.05, 1.0))
    def run(n, cells, seed, kind, threads, share, prefix, slack, pool_frac):
        rng = np.random.default_rng(seed)
        p = {"flat": np.ones(n), "ramp": np.linspace(1.0, 0.05, n), "spiky": rng.random(n) ** 12 + 1e-12,
             "zeros": np.where(rng.random(n) < 0.4, 0.0, rng.random(n) + 1e-3)}[kind]
        if not (p > 0).any():
            p[0] = 1.0
        p = p / p.sum()
        size = int(rng.integers(1, int((p > 0).sum()) + 1))
        for k, v in (("VCY_CHOICE_THREADS", threads), ("VCY_CHOICE_MIN_SHARE", share), ("VCY_CHOICE_PREFIX", prefix), ("VCY_CHOICE_SLACK", slack)):
            monkeypatch.setenv(k, str(v))
        rs = np.random.RandomState(seed % (2 ** 32))
        want = np.stack([rs.choice(n, size=(size,), replace=False, p=p) for _ in range(cells)], 0)
        big = np.random.RandomState(seed % (2 ** 32)).random_sample(cells * size * 12 + 64)
        out = np.full((cells, size), -1, dtype=np.int64)
        cd, used = ctypes.c_int64(0), ctypes.c_int64(0)
        assert L.vcy_choice_stream_host(big.ctypes.data, big.size, p.ctypes.data, n, size, cells, out.ctypes.data, ctypes.byref(cd), ctypes.byref(used)) == 0
        assert cd.value == cells and np.array_equal(out, want)
        check = np.random.RandomState(seed % (2 ** 32))
        check.random_sample(used.value)
        assert np.array_equal(check.random_sample(3), rs.random_sample(3))
        cut = int(used.value * pool_frac)
        out2 = np.full((cells, size), -1, dtype=np.int64)
        cd2, used2 = ctypes.c_int64(0), ctypes.c_int64(0)
        assert L.vcy_choice_stream_host(big.ctypes.data, cut, p.ctypes.data, n, size, cells, out2.ctypes.data, ctypes.byref(cd2), ctypes.byref(used2)) == 0
        assert cd2.value <= cells and used2.value <= cut and np.array_equal(out2[: cd2.value], want[: cd2.value])
        if cd2.value < cells:                                                      # the sequential replay stops at the same cell
            monkeypatch.setenv("VCY_CHOICE_THREADS", "1")
            cd3, used3 = ctypes.c_int64(0), ctypes.c_int64(0)
            L.vcy_choice_stream_host(big.ctypes.data, cut, p.ctypes.data, n, size, cells, out2.ctypes.data, ctypes.byref(cd3), ctypes.byref(used3))
            assert (cd3.value, used3.value) == (cd2.value, used2.value)

    run()


def test_atlas_memory_plan_for_one_million_cells():
    """The per-rank memory plan of the atlas path is plain arithmetic: 1M cells x 30k genes on 8 ranks must fit
    288 GB per rank with room to spare in resident mode, and a streamed single rank must stay O(block)."""
    import velocyto_amd
    from velocyto_amd import atlas
    plan = atlas.memory_plan(1_000_000, 30_000, 2400, 8, 0, nrndm=250, k=30, count_bytes=1, halo_e=0.1, halo_k=0.25)
    assert plan["cells_per_rank"] == 125_000
    assert 2.5 < plan["csr_layers_with_halo_GB"] < 4.5 and 28 < plan["block_Sx_Ux_GB"] < 34
    assert plan["total_GB"] < 0.3 * 288
    one = atlas.memory_plan(1_000_000, 30_000, 2400, 1, 50_000, nrndm=250, k=30, count_bytes=1)
    assert one["block_Sx_Ux_GB"] < 20 and one["csr_layers_with_halo_GB"] < 40 and one["total_GB"] < 288
    dense = 1_000_000 * 30_016 * 4 * 2 / 1e9
    assert one["total_GB"] < 0.5 * dense                   # against Sx + Ux resident (240 GB)


def test_dpp_operands_of_the_scaling_kernel_are_not_fresh_valu_results(tmp_path):
    """k_embedding_scaling<double> reads a member's weights through `v_fmac_f64_dpp ... row_newbcast` written as inline asm
    (csrc/scaling.hip: fmac_bcast).  gfx9 needs two wait states between a VALU write of a VGPR and a DPP read of it, and the
    compiler's hazard recogniser does not look inside an asm statement: the listing must not hold a VALU instruction that writes
    the DPP operand within the two instructions before it (the weights come straight from an LDS read)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc here")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "scaling.s"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I" + os.path.join(root, "include"),
                        os.path.join(root, "velocyto.py_amd", "csrc", "scaling.hip"), "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]

    def regs(tok):
        m = re.match(r"v\[(\d+):(\d+)\]", tok)
        if m:
            return set(range(int(m.group(1)), int(m.group(2)) + 1))
        m = re.match(r"v(\d+)$", tok)
        return {int(m.group(1))} if m else set()

    instrs = [l.strip() for l in out.read_text().split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    seen = 0
    for i, l in enumerate(instrs):
        if not l.startswith("v_fmac_f64_dpp"):
            continue
        seen += 1
        src0 = regs([o.strip() for o in l.split(None, 1)[1].split(",")][1])
        for back in (1, 2):
            p = instrs[i - back]
            if p.startswith("v_") and not p.startswith("v_cmp"):
                dst = p.split(None, 1)[1].split(",")[0].strip()
                assert not (regs(dst) & src0), f"{p}  ->  {l}"
    assert seen >= 96                                      # 8 members x 2 elements x (1 + 2) weight sets x 3 rows in flight, at least
