"""The cell-sharded pipeline of bench.py with MORE THAN ONE RANK on a one-GPU box: every rank runs on cuda:0 and the
collectives go through gloo (host-staged; `VCY_SINGLE_DEVICE=1 VCY_DIST_BACKEND=gloo`), so that everything except the
RCCL transport itself is the code the 2/4/8-GPU runs execute: Morton relabelling, shard bounds, kNN queries of a shard
against all cells, pooling of a shard, all-reduce of the fit moments, halo plan with SHARDED e (a compact own + halo
buffer and renumbered neighbour lists; `--exchange allgather` keeps the full-height buffer with cell0 / u_row0 offsets),
stage D split into the interior cells (run while the halo moves) and the cells with remote neighbours, all-gather of the
correlation rows.

`test_rccl_collectives_on_one_gpu` runs the same code on the REAL transport: backend "nccl" (= RCCL) at world size 1 with
the collectives forced, launched through bench.py's own self-launcher, and must reproduce the plain one-rank run.

The sharded results must equal the one-rank run of the same (relabelled) problem: neighbour samples and labels
identical, gamma to fp64-summation-order tolerance, correlations to 2e-6.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--no-cpu-baseline", "--no-extra", "--cells", "4100", "--genes", "1536", "--n-neighbors", "100", "--k", "12", "--steps", "1", "--warmup", "1"]


def run(world, dump, extra=(), port=29611, backend="gloo", force="1", self_launch=False, env_extra=None):
    env = dict(os.environ, VCY_SINGLE_DEVICE="1", VCY_DIST_BACKEND=backend, VCY_FORCE_COLLECTIVES=force, MASTER_PORT=str(port),
               MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    if world == 1 or self_launch:            # self_launch: bench.py spawns its own ranks (no torch.distributed.run wrapper)
        cmd = [sys.executable, "bench.py", "--gpus", str(world), *ARGS, "--dump", dump, *extra]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), "bench.py", "--gpus", str(world), *ARGS, "--dump", dump, *extra]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    last = [l for l in r.stdout.strip().splitlines() if l.strip()][-1]
    assert last.startswith("{") and '"n_gpus": %d' % world in last, last[:200]      # the JSON line is the last thing on stdout
    return dict(np.load(dump)), last


@pytest.mark.parametrize("world,extra", [(2, ()), (3, ("--exchange", "allgather")), (3, ()), (2, ("--no-overlap",)), (2, ("--no-fuse",))])
def test_sharded_pipeline_equals_one_rank(tmp_path, world, extra):
    from velocyto_amd import ops
    ops.require_gpu()
    one, _ = run(1, str(tmp_path / "one.npz"), extra, port=29611 + world)
    many, _ = run(world, str(tmp_path / "many.npz"), extra, port=29631 + world + len(extra))
    assert np.array_equal(one["perm"], many["perm"]) and np.array_equal(one["neigh"], many["neigh"])
    np.testing.assert_allclose(many["gamma"], one["gamma"], rtol=2e-6, atol=1e-9)
    fin = np.isfinite(one["corr"])
    assert np.array_equal(np.isfinite(many["corr"]), fin)
    np.testing.assert_allclose(many["corr"][fin], one["corr"][fin], atol=2e-6)
    assert one["corr"].shape == (4100, 50) and fin.mean() > 0.99


def test_failed_collective_self_check_falls_back_to_allgather(tmp_path):
    """bench.py's start-up self-check (distributed.self_check): the uneven all-to-all made to fail on rank 1 only -> every rank
    reports it by name, the run switches to --exchange allgather on all ranks and still reproduces the one-rank results."""
    import json
    from velocyto_amd import ops
    ops.require_gpu()
    one, j1 = run(1, str(tmp_path / "one.npz"), port=29671)
    many, j2 = run(2, str(tmp_path / "two.npz"), port=29672, env_extra={"VCY_SELF_CHECK_FAIL": "all_to_all_uneven@1"})
    chk1, chk2 = json.loads(j1)["collective_self_check"], json.loads(j2)["collective_self_check"]
    assert chk1.startswith("all ok") and "all_to_all_uneven" in chk1                 # forced collectives at one rank: checks ran
    assert chk2.startswith("FAILED: all_to_all_uneven") and chk2.endswith("-> --exchange allgather")
    assert "all-gather of Sx shards" in json.loads(j2)["config"]["parallelism"]
    assert np.array_equal(one["neigh"], many["neigh"])
    fin = np.isfinite(one["corr"])
    np.testing.assert_allclose(many["corr"][fin], one["corr"][fin], atol=2e-6)


def test_rccl_collectives_on_one_gpu(tmp_path):
    """init_process_group("nccl"), all_reduce, all_gather_into_tensor, all_to_all_single (halo) and barrier on RCCL - the
    transport the multi-GPU runs use - at world size 1 with the collectives forced, against the same relabelled problem
    run over gloo (which the tests above tie to the 2- and 3-rank runs)."""
    import json
    from velocyto_amd import ops
    ops.require_gpu()
    ref, line0 = run(1, str(tmp_path / "gloo.npz"), port=29701)
    rccl, line1 = run(1, str(tmp_path / "rccl.npz"), port=29702, backend="nccl")
    agat, line2 = run(1, str(tmp_path / "agat.npz"), ("--exchange", "allgather"), port=29703, backend="nccl")
    assert json.loads(line0)["rccl_ranks"] == 0 and json.loads(line1)["rccl_ranks"] == 1 and json.loads(line2)["rccl_ranks"] == 1
    for got in (rccl, agat):
        assert np.array_equal(got["perm"], ref["perm"]) and np.array_equal(got["neigh"], ref["neigh"])
        np.testing.assert_array_equal(got["gamma"], ref["gamma"])
        fin = np.isfinite(ref["corr"])
        assert np.array_equal(np.isfinite(got["corr"]), fin)
        if got is rccl:                      # same schedules, same launches: the transport must not change a bit
            np.testing.assert_array_equal(got["corr"][fin], ref["corr"][fin])
        else:                                # the all-gather exchange runs stage D as ONE launch, the halo exchange as whole rounds + the rest
            np.testing.assert_allclose(got["corr"][fin], ref["corr"][fin], atol=2e-6)     # (other tiles / kernels for the last cells)


def test_bench_self_launch_equals_torchrun(tmp_path):
    """`python bench.py --gpus 2` without a launcher spawns one process per rank itself (torch.multiprocessing) and must give
    what the torch.distributed.run launch gives."""
    from velocyto_amd import ops
    ops.require_gpu()
    a, _ = run(2, str(tmp_path / "torchrun.npz"), port=29751)
    b, line = run(2, str(tmp_path / "self.npz"), port=29752, self_launch=True)
    assert '"n_gpus": 2' in line
    assert np.array_equal(a["neigh"], b["neigh"]) and np.array_equal(a["gamma"], b["gamma"])
    fin = np.isfinite(a["corr"])
    assert np.array_equal(np.isfinite(b["corr"]), fin) and np.array_equal(b["corr"][fin], a["corr"][fin])


def test_eight_ranks_at_full_size_on_one_device(tmp_path):
    """BASELINE.json configs[3] rehearsed on one device: the 50 000 x 30 000 problem of the headline cut into EIGHT shards of 6 250
    cells, one process per shard, every process on cuda:0 with the collectives host-staged through gloo - everything of the
    8-GPU run except the transport: relabelling along the Hilbert curve, shard-sized kNN queries and pooling, the all-reduce of
    the fit moments, the halo plan with sharded e (compact own + halo buffers, renumbered lists, zero-length splits between
    ranks that share no row), stage D split into interior cells and cells with remote neighbours, the all-gather of the
    correlation rows, the whole-matrix decision on the branch rule.  Labels, neighbour samples, gamma and ALL 12.5 M
    correlation rows must equal the one-rank run; the JSON line must carry every rank's own account of the pass."""
    import json
    from velocyto_amd import ops
    ops.require_gpu()
    if torch.cuda.mem_get_info()[1] < 200e9:
        pytest.skip("needs the memory of one MI355X for eight co-resident ranks")
    full = ["--no-cpu-baseline", "--no-extra", "--steps", "1", "--warmup", "1"]

    def go(world, dump, port):
        env = dict(os.environ, VCY_SINGLE_DEVICE="1", VCY_DIST_BACKEND="gloo", VCY_FORCE_COLLECTIVES="1", MASTER_PORT=str(port),
                   MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        r = subprocess.run([sys.executable, "bench.py", "--gpus", str(world), *full, "--dump", dump], cwd=ROOT, env=env, capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        return dict(np.load(dump)), json.loads([l for l in r.stdout.strip().splitlines() if l.strip()][-1])
    one, j1 = go(1, str(tmp_path / "one.npz"), 29801)
    many, j8 = go(8, str(tmp_path / "eight.npz"), 29808)
    assert j8["n_gpus"] == 8 and j8["config"]["cells"] == 50000 and j8["config"]["genes"] == 30000 and j8["config"]["nrndm"] == 250
    assert np.array_equal(one["perm"], many["perm"]) and np.array_equal(one["neigh"], many["neigh"])
    np.testing.assert_allclose(many["gamma"], one["gamma"], rtol=2e-6, atol=1e-9)
    assert one["corr"].shape == (50000, 250)
    fin = np.isfinite(one["corr"])
    assert np.array_equal(np.isfinite(many["corr"]), fin) and fin.mean() > 0.999
    assert np.abs(many["corr"][fin] - one["corr"][fin]).max() <= 2e-6
    det = j8["config"]["parallelism_detail"]
    ranks = det["per_rank"]
    assert len(ranks) == 8 and sum(r["cells"] for r in ranks) == 50000 and all(r["cells"] == 6250 for r in ranks)
    assert all(0 < r["halo_rows_received"] < 50000 - 6250 for r in ranks)             # a halo, not the whole matrix
    assert sum(r["halo_rows_received"] for r in ranks) == sum(r["halo_rows_sent"] for r in ranks)
    assert all(0 < r["interior_cells"] < 6250 for r in ranks) and all(r["stage_ms"]["D_coldeltacor"] > 0 for r in ranks)
    # the launch that overlaps the transfer holds whole device rounds of interior cells (distributed.overlap_schedules)
    per_round = 256 * (6 if j8["dtype"] == "f64" else 8)        # one workgroup per CU, 6 (f64) / 8 (f32) cells per workgroup
    assert all(r["cells_run_while_the_halo_moves"] % per_round == 0 and r["cells_run_while_the_halo_moves"] <= r["interior_cells"] for r in ranks)
    b = det["bytes_per_collective"]
    es = {"f32": 4, "f64": 8}[j8["dtype"]]                  # the bench's default arithmetic is the reference's (f64)
    assert j8["dtype"] == j1["dtype"] == "f64"
    assert b["B_all_reduce_fit_moments"] == 3 * 30000 * 8 and b["D_all_gather_correlation_rows_total"] == 50000 * 250 * es
    assert b["D_halo_all_to_all_received_per_rank"] == [r["halo_rows_received"] * 30016 * es for r in ranks]
    assert j8["config"]["stage_D_rule"] == j1["config"]["stage_D_rule"]
