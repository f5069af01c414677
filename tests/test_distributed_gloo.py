"""World-size-2 (and 3, ragged) checks of the cell-sharding logic over torch.distributed `gloo` on CPU.

The collectives of the N>1 path (velocyto_amd.distributed) are exercised with the CPU oracle standing
in for the per-shard kernels: sharded stage B (all-reduce of per-gene moments) and stage D
(all-gather of Sx shards, then of compact correlation rows) must reproduce the unsharded result.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, C, G, nr, q):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import velocyto_amd
        from velocyto_amd import distributed as D
        import oracle
        rng = np.random.default_rng(7)                       # identical replicated inputs on every rank
        Sx = rng.gamma(2.0, 1.0, (G, C))
        Ux = rng.gamma(1.0, 1.0, (G, C)) * (rng.random((G, C)) < 0.7)
        dmat = rng.normal(size=(G, C))
        ixs = np.stack([rng.choice(C, nr, replace=False) for _ in range(C)])
        c0, c1 = D.shard_bounds(C, world, rank)
        assert D.world() == (rank, world)
        # B: local moments over my cells -> all-reduce -> gamma
        mom = torch.tensor(np.stack([(Sx[:, c0:c1] ** 2).sum(1), (Sx[:, c0:c1] * Ux[:, c0:c1]).sum(1), (Ux[:, c0:c1] ** 2).sum(1)]))
        D.all_reduce_sum(mom)
        gamma = np.maximum(0, mom[1].numpy() / mom[0].numpy()).astype(np.float32)
        # D: all-gather the Sx shards (cells-major rows), correlations of my cells, all-gather the compact rows
        Sx_loc = torch.tensor(np.ascontiguousarray(Sx.T[c0:c1]))
        Sx_full = D.all_gather_rows(Sx_loc, C).numpy().T
        corr_loc = oracle.coldeltacor_partial_compact(Sx_full, dmat, ixs, "sqrt", 1e-10, c0=c0, c1=c1)[c0:c1]
        corr = D.all_gather_rows(torch.tensor(corr_loc), C)
        out_buf = torch.empty((C, nr), dtype=torch.float64)
        assert D.all_gather_rows(torch.tensor(corr_loc), C, out=out_buf) is out_buf and torch.equal(torch.nan_to_num(out_buf, nan=9.0), torch.nan_to_num(corr, nan=9.0))
        # halo exchange: only the rows my neighbour lists reference travel; results must not change
        need = torch.zeros(C, dtype=torch.bool)
        need[torch.as_tensor(ixs[c0:c1].ravel())] = True
        need[c0:c1] = True
        plan = D.HaloPlan(need, C)
        full2 = torch.full((C, G), float("nan"), dtype=torch.float64)
        plan.exchange(Sx_loc, full2)
        got = full2.numpy()
        assert np.array_equal(got[need.numpy()], Sx.T[need.numpy()]), "halo rows differ"
        assert np.isnan(got[~need.numpy()]).all(), "rows nobody asked for must stay untouched"
        assert plan.n_recv == int(need.sum()) - (c1 - c0)
        plan.exchange(Sx_loc * 2.0, full2)                                   # plan is reusable
        assert np.array_equal(full2.numpy()[need.numpy()], 2.0 * Sx.T[need.numpy()])
        corr_h = oracle.coldeltacor_partial_compact(np.nan_to_num(full2.numpy().T / 2.0), dmat, ixs, "sqrt", 1e-10, c0=c0, c1=c1)[c0:c1]
        assert np.array_equal(np.nan_to_num(corr_h, nan=9.0), np.nan_to_num(corr_loc, nan=9.0))
        # SHARDED e: own rows + halo rows in a compact buffer, neighbour lists renumbered (no full-height copy of e anywhere)
        n_loc = c1 - c0
        compact = torch.full((n_loc + plan.n_recv, G), float("nan"), dtype=torch.float64)
        compact[:n_loc] = Sx_loc
        h = plan.begin(Sx_loc, recv_out=compact[n_loc:])
        plan.end(h, compact, row0=n_loc)
        loc = plan.localize(torch.as_tensor(ixs[c0:c1]))
        assert int(loc.min()) >= 0 and int(loc.max()) < compact.shape[0] and not bool(torch.isnan(compact).any())
        assert torch.equal(compact[loc.long()], torch.as_tensor(Sx.T)[torch.as_tensor(ixs[c0:c1])]), "renumbered lists must address the same rows"
        # stage D on the compact buffer: cells 0..n_loc-1 of a (n_loc + n_halo)-cell problem, d holds the own rows only
        d_pad = np.zeros((G, compact.shape[0]))
        d_pad[:, :n_loc] = dmat[:, c0:c1]
        corr_c = oracle.coldeltacor_partial_compact(compact.numpy().T.copy(), d_pad, np.vstack([loc.numpy(), np.zeros((plan.n_recv, nr), dtype=np.int32)]),
                                                    "sqrt", 1e-10, c0=0, c1=n_loc)[:n_loc]
        assert np.array_equal(np.nan_to_num(corr_c, nan=9.0), np.nan_to_num(corr_loc, nan=9.0)), "sharded e changed the correlations"
        if rank == 0:
            q.put((gamma, corr.numpy(), Sx_full))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,C", [(2, 40), (2, 41), (3, 41)])
def test_sharded_path_matches_unsharded(world, C, oracle):
    G, nr = 50, 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, C, G, nr, q)) for r in range(world)]
    for p in procs:
        p.start()
    gamma, corr, Sx_full = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(7)
    Sx = rng.gamma(2.0, 1.0, (G, C))
    Ux = rng.gamma(1.0, 1.0, (G, C)) * (rng.random((G, C)) < 0.7)
    dmat = rng.normal(size=(G, C))
    ixs = np.stack([rng.choice(C, nr, replace=False) for _ in range(C)])
    np.testing.assert_array_equal(Sx_full, Sx)
    np.testing.assert_allclose(gamma, oracle.fit_slope(Ux, Sx), rtol=1e-6)
    ref = oracle.coldeltacor_partial_compact(Sx, dmat, ixs, "sqrt", 1e-10)
    np.testing.assert_array_equal(np.isnan(corr), np.isnan(ref))
    np.testing.assert_allclose(np.nan_to_num(corr), np.nan_to_num(ref), atol=1e-14)


def test_shard_bounds_cover_and_balance():
    sys.path.insert(0, ROOT)
    import velocyto_amd
    from velocyto_amd import distributed as D
    for n in (1, 7, 8, 50000, 50001):
        for w in (1, 2, 3, 8):
            b = D.all_shard_bounds(n, w)
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [y - x for x, y in b]
            assert max(sizes) - min(sizes) <= 1
    assert D.world() == (0, 1)
    t = torch.arange(6.0).reshape(3, 2)
    assert D.all_gather_rows(t, 3) is t and D.all_reduce_sum(t) is t       # world size 1: no-ops


def _ragged_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import velocyto_amd
        from velocyto_amd import distributed as D
        rng = np.random.default_rng(11)                       # the same "dataset" on every rank: CSR rows of 37 cells
        C, G = 37, 50
        lens_all = rng.integers(0, 9, C)
        lens_all[5] = 0                                       # an empty row travels as a zero-length segment
        ptr = np.concatenate([[0], np.cumsum(lens_all)])
        idx_all = np.concatenate([np.sort(rng.choice(G, n, replace=False)) for n in lens_all]).astype(np.int32)
        dat_all = rng.integers(1, 200, idx_all.size).astype(np.uint8)
        c0, c1 = D.shard_bounds(C, world, rank)
        need = torch.zeros(C, dtype=torch.bool)
        need[torch.as_tensor(rng.choice(C, 12, replace=False))] = True      # same draw on every rank, then make it rank-specific
        need = torch.roll(need, 5 * rank)
        need[c0:c1] = True
        plan = D.HaloPlan(need, C)
        lens = torch.as_tensor(lens_all[c0:c1])
        own_idx, own_dat = torch.as_tensor(idx_all[ptr[c0]:ptr[c1]]), torch.as_tensor(dat_all[ptr[c0]:ptr[c1]])
        lens_h, idx_h, dat_h = plan.fetch_ragged(lens, own_idx, own_dat)
        rows = plan.recv_idx.numpy()                          # global numbers of the halo rows, ascending
        assert np.array_equal(rows, np.nonzero(need.numpy() & ~np.isin(np.arange(C), np.arange(c0, c1)))[0])
        assert np.array_equal(lens_h.numpy(), lens_all[rows])
        assert np.array_equal(idx_h.numpy(), np.concatenate([idx_all[ptr[r]:ptr[r + 1]] for r in rows] + [np.empty(0, np.int32)]))
        assert np.array_equal(dat_h.numpy(), np.concatenate([dat_all[ptr[r]:ptr[r + 1]] for r in rows] + [np.empty(0, np.uint8)]))
        # small per-row vectors (graph rows, size factors) ride the same plan
        f = torch.as_tensor(np.arange(C, dtype=np.float64)[c0:c1] * 1.5)
        assert np.array_equal(plan.fetch(f.reshape(-1, 1)).reshape(-1).numpy(), rows * 1.5)
        # and the renumbering into [own | halo]
        some = torch.as_tensor(np.concatenate([rows[:3], np.arange(c0, min(c1, c0 + 2))]))
        loc = plan.localize(some)
        exp = np.concatenate([(c1 - c0) + np.arange(min(3, rows.size)), np.arange(min(c1 - c0, 2))])
        assert np.array_equal(loc.numpy(), exp)
        q.put(rank)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ragged_count_row_halo(world):
    """The count-row halo of the atlas path (atlas.py): ragged CSR rows, row lengths and per-row vectors travel through one
    HaloPlan (all-to-all with uneven splits; gloo point-to-point here) and arrive in ascending global row order."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ragged_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    done = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert done == list(range(world))


def test_overlap_schedules_round_alignment():
    """distributed.overlap_schedules: the first launch holds whole rounds of interior cells in schedule order, the second everything
    else in schedule order; together they are a permutation of the rank's cells."""
    import torch
    from velocyto_amd import distributed as D
    g = torch.Generator().manual_seed(3)
    n = 6250
    base = torch.randperm(n, generator=g)
    interior = torch.rand(n, generator=g) < 0.6
    first, second = D.overlap_schedules(base, interior, 2048)
    assert first.numel() == (int(interior.sum()) // 2048) * 2048 and first.numel() + second.numel() == n
    assert bool(interior[first.long()].all())
    assert torch.equal(torch.sort(torch.cat([first, second]).long()).values, torch.arange(n))
    pos = torch.empty(n, dtype=torch.long); pos[base] = torch.arange(n)
    assert bool((pos[first.long()][1:] > pos[first.long()][:-1]).all()) and bool((pos[second.long()][1:] > pos[second.long()][:-1]).all())
    # fewer interior cells than one round: nothing runs before the halo has landed
    first, second = D.overlap_schedules(base, torch.zeros(n, dtype=torch.bool).index_fill_(0, base[:100], True), 2048)
    assert first.numel() == 0 and torch.equal(second.long(), base)


def _self_check_worker(rank, world, port, inject, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), VCY_SELF_CHECK_FAIL=inject)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import velocyto_amd
        from velocyto_amd import distributed as D
        q.put((rank, D.self_check(torch.device("cpu"))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,inject", [(2, ""), (4, ""), (3, "all_to_all_uneven@1"), (8, ""), (3, "all_to_all_halo_sizes@2")])      # 4 ranks: empty segments between different ranks; 8: the split table of cfg3
def test_collective_self_check(world, inject):
    """distributed.self_check: every collective shape of the sharded path on tiny tensors (uneven all-to-all with empty
    segments, ragged and equal all-gathers, SUM / MIN all-reduces, the halo masks).  A failure on ONE rank becomes the same
    verdict on every rank (bench.py then switches all of them to --exchange allgather)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_self_check_worker, args=(r, world, port, inject, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from velocyto_amd import distributed as D
    for rank, res in got.items():
        assert set(res["agreed"]) == set(D.SELF_CHECKS)
        if not inject:
            assert all(res[n] == "ok" for n in D.SELF_CHECKS), res
            assert all(res["agreed"].values())
        else:
            name, bad_rank = inject.split("@")
            assert (res[name] != "ok") == (rank == int(bad_rank))              # only that rank saw it fail ...
            assert res["agreed"][name] is False                                # ... every rank knows
            assert all(ok for n, ok in res["agreed"].items() if n != name)


def _disagree_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), VCY_SELF_CHECK_FITS="0,3,5")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import velocyto_amd
        from velocyto_amd import distributed as D
        res = D.self_check(torch.device("cpu"))
        m = D.all_reduce_max(torch.tensor([float(rank) * 1.5], dtype=torch.float64))
        q.put((rank, res, float(m)))
    finally:
        dist.destroy_process_group()


def test_self_check_row_width_is_a_collective_decision():
    """The halo-size check picks its row width (30 016 fp64 columns when the device has room, 4 fp32 otherwise) from a LOCAL fact - free
    memory.  Ranks that disagreed would enter all_to_all_single with different element sizes (a hang or corruption on RCCL): the choice is
    all-reduced (MIN).  Here three of eight ranks claim the wide buffers fit; every rank must run the narrow check and pass it.  Also
    distributed.all_reduce_max, the reduction the atlas path judges its fp64 sqrt domain with before any rank raises."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_disagree_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, res, m in got:
        assert res["all_to_all_halo_sizes"] == "ok" and all(res["agreed"].values()), (rank, res)
        assert m == 1.5 * (world - 1)
