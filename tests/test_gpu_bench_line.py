"""The ONE JSON line of bench.py, with everything SURVEY.md 8(d) asks for, on a small instance of the workload: the headline in
the reference's arithmetic (f64, literal rule) with its roofline block, the narrower production modes beside it, the extra
lines (E, F, default fit_gammas, the reference-default list width, the randomised control, cfg2) and the CPU baseline with
its parity block.  The full-size numbers are the driver's; this test pins the STRUCTURE the driver's record is parsed from
and that no guarded line fails."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_structure():
    from velocyto_amd import ops
    ops.require_gpu()
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "bench.py", "--cells", "6000", "--genes", "2048", "--n-neighbors", "100", "--k", "12", "--steps", "2", "--warmup", "1",
           "--cpu-cells", "128", "--extra-budget-s", "300"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.strip()]
    j = json.loads(lines[-1])
    # ---- the headline: the reference's arithmetic, timed with the flags given
    assert j["dtype"] == "f64" and j["steps"] == 2 and j["warmup"] == 1 and j["n_gpus"] == 1 and j["unit"] == "cells/s"
    assert abs(j["value"] - 6000 / (j["ms_per_step"] * 1e-3)) < 1e-6 * j["value"]
    assert "literal" in j["config"]["stage_D_rule"] and j["config"]["arithmetic"].startswith("f64")
    roof = j["roofline"]
    assert roof["dtype"] == "f64" and "double" in roof["kernel"] and roof["bound"] == "valu" and roof["avg_launch_ms"] > 0
    assert 0.3 < roof["effective_clock_ghz"] < 3.0 and roof["effective_clock"]["launches"] == 3          # measured in this run: three extra, untimed steps
    assert not any(k in roof for k in ("wave_time",))                               # what comes from a committed profile is named profile_*
    assert j["config"]["count_layer_dtype"] == "uint16"                             # the loom's type is what `value` ran on
    for k in ("A_knn_search_ms", "A_pooling_ms", "B_fit_slope_ms", "D_coldeltacor_ms"):
        assert j["config"][k] > 0
    # ---- the production modes beside it, each with its own roofline block and its distance from the headline
    pm = j["precision_modes"]
    assert "not_measured" not in pm, pm.get("not_measured")
    f32 = pm["f32_production"]
    assert f32["dtype"] == "f32" and f32["roofline"]["dtype"] == "f32" and "float" in f32["roofline"]["kernel"]
    assert f32["vs_headline"]["max_abs_dcorr_all_pairs"] < 5e-5 and f32["vs_headline"]["nan_pattern_equal"]
    assert f32["cells_per_s"] > j["value"]                                  # narrower arithmetic is faster - and is not the headline
    assert pm["headline"]["dtype"] == "f64"
    u8 = pm["f64_uint8_layers"]            # the survey generator's spliced layer has counts above 255: the layers do not narrow, and the line says so
    narrows = "skipped" not in u8
    if narrows:
        assert u8["steps"] == 2 and u8["same_results_as_uint16"] and u8["cells_per_s"] > 0
    else:
        assert "do not narrow" in u8["skipped"]
    assert "survey" in j["config"]["generator"]
    tel = j["telemetry"]
    assert set(tel) >= {"smi_before_timed_steps", "smi_after_timed_steps"} and len(roof["effective_clock"]["per_xcd_mean"]) == 8
    # ---- SURVEY 8(d)'s other lines, all present and none of them failed or skipped
    ex = j["extra"]
    for name in ("randomised_control", "D_reference_defaults_nrndm3000", "facade", "cfg2"):
        assert name in ex and "error" not in ex[name] and "skipped" not in ex[name], (name, ex.get(name))
    fac = ex["facade"]
    for k in ("B_fit_gammas_default_ms", "E_calculate_embedding_shift_ms", "F_prepare_markov_ms", "F_run_markov_ms", "F_run_markov_ms_per_step",
              "A_knn_imputation_ms", "D_estimate_transition_prob_ms"):
        assert fac[k] > 0, k
    assert fac["dtype"] == "f64" and fac["F_run_markov_steps"] == 2500
    wide = ex["D_reference_defaults_nrndm3000"]
    assert wide["n_neighbors"] == 1200 and wide["nrndm"] == int(0.3 * 1201) and wide["finite_fraction"] > 0.99 and wide["max_abs_corr"] <= 1.0 + 1e-9
    assert wide["roofline"]["bound"] == "valu" and wide["roofline"]["mix_floor_ms"] > 0 and "128 whole cells" in wide["parity"]
    c2 = ex["cfg2"]
    assert (c2["cells"], c2["genes"]) == (10000, 20000)
    for name in ("unbalanced", "balanced"):
        assert c2[name]["A_knn_imputation_ms"] > 0 and c2[name]["B_fit_slope_ms"] > 0
    assert ex["randomised_control"]["dual_over_single"] < 2.0
    # ---- the same numbers as scalars of `config` (what the driver's record keeps)
    cfg = j["config"]
    for k in ("E_calculate_embedding_shift_ms", "F_run_markov_ms_per_step", "B_fit_gammas_default_ms", "D_reference_defaults_nrndm3000_ms",
              "cfg2_unbalanced_A_ms", "cfg2_unbalanced_B_ms", "cfg2_balanced_A_ms", "cfg2_balanced_B_ms", "f32_production_cells_per_s", "D_dual_control_over_single",
              *(("f64_uint8_layers_cells_per_s",) if narrows else ())):
        assert isinstance(cfg[k], float) and cfg[k] > 0, k
    # ---- CPU baseline: stage D at full width (the run's own matrices and lists) by the reference's kernel and the restatement, the closed
    #      sub-problem for stages A - C and the HIP path's distance from the oracle on the same inputs
    cb = j["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["stage_D_cells_per_s"] >= cb["value"]
    full, closed = cb["full_width"], cb["closed_subproblem"]
    assert "error" not in full and "skipped" not in full, full
    assert full["e_stride_cells"] == 6000 and "stride 6000 cells" in cb["sample"]
    for r in full["restatement"].values():
        assert r["cells_per_s"] > 0 and r["nan_pattern_equal"] and r["max_abs_dcorr_hip_vs_restatement"] < 1e-9
    if cb["kind"] == "reference":          # oracle/_ref travelled with the snapshot: stage D by the reference's own Cython kernel
        assert full["reference_kernel"]["cells_per_s"] > 0 and full["reference_kernel"]["threads"] == cb["cores"]
        rk = closed["reference_kernel"]
        assert rk["D_s"] > 0 and rk["nan_pattern_equal"] and rk["max_abs_dcorr_restatement_vs_reference"] < 1e-12
    assert cb["parity"]["f64"]["max_abs_dcorr"] < 1e-8 and cb["parity"]["f32_nopsc"]["max_abs_dcorr"] < 5e-5
    same = cb["parity"]["f64"]["stage_D_on_identical_inputs"]                       # stage D alone on the HIP path's own pooled matrices
    assert same["max_abs_dcorr"] < 1e-10 and same["nan_pattern_equal"]
