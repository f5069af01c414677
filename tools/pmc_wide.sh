#!/bin/bash
# on the GPU box: counters of the stage-D launch at the reference's default list width (nrndm = 3000) -> profiles/<tag>_cdc_wide_counters.json
# usage: tools/pmc_wide.sh <tag>      (separate --pmc passes, never combined with API tracing)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=${1:-r06}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pw_*
CMD="python $R/tools/bench_wide.py"
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS \
    --kernel-trace --output-format csv -d /tmp/pw_sq -- $CMD > /tmp/pw_sq.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pw_fetch -- $CMD > /tmp/pw_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw_write -- $CMD > /tmp/pw_write.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pw_grbm -- $CMD > /tmp/pw_grbm.log 2>&1
tail -1 /tmp/pw_sq.log | cut -c1-400
python - "$R" "$T" <<'PY'
import csv, glob, json, sys, collections
R, T = sys.argv[1], sys.argv[2]
def longest(d):
    """counters of the longest dispatch of the f64 grouped kernel in one pass: {counter: value summed over its rows}, duration ms"""
    rows = []
    for f in glob.glob(f"/tmp/pw_{d}/**/*counter_collection.csv", recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if "k_cdc_partial_grouped<double" in r["Kernel_Name"]]
    by = collections.defaultdict(list)
    for r in rows:
        by[r["Dispatch_Id"]].append(r)
    best = max(by.values(), key=lambda rs: int(rs[0]["End_Timestamp"]) - int(rs[0]["Start_Timestamp"]))
    out = collections.defaultdict(float)
    for r in best:
        out[r["Counter_Name"]] += float(r["Counter_Value"])
    return dict(out), (int(best[0]["End_Timestamp"]) - int(best[0]["Start_Timestamp"])) / 1e6, best[0]["Kernel_Name"][:90]
sq, ms, name = longest("sq")
fetch, _, _ = longest("fetch")
write, _, _ = longest("write")
grbm, gms, _ = longest("grbm")
C, G, nr, chunk = 50000, 30000, 3000, 1024
pair_chunks = C * nr * ((G + chunk - 1) // chunk)
waves = sq.get("SQ_WAVE_CYCLES", 0.0)
rec = {"f64_wide": {
    "kernel": name, "rules": 1, "workload": {"cells": C, "genes": G, "nrndm": nr, "genes_per_chunk": chunk, "pair_chunks_per_launch": pair_chunks},
    "SQ_INSTS_VALU_per_launch": sq.get("SQ_INSTS_VALU"), "valu_insts_per_pair_chunk": sq.get("SQ_INSTS_VALU", 0.0) / pair_chunks,
    "SQ_INSTS_SALU_per_launch": sq.get("SQ_INSTS_SALU"), "SQ_INSTS_LDS_per_launch": sq.get("SQ_INSTS_LDS"),
    "FETCH_SIZE_KiB": fetch.get("FETCH_SIZE"), "WRITE_SIZE_KiB": write.get("WRITE_SIZE"),
    "hbm_bytes_per_launch": 2.0 * fetch.get("FETCH_SIZE", 0.0) * 1024 + write.get("WRITE_SIZE", 0.0) * 1024,
    "profiled_launch_ms": ms, "GRBM_GUI_ACTIVE_per_launch": grbm.get("GRBM_GUI_ACTIVE"), "grbm_pass_launch_ms": gms,
    "effective_clock_ghz": grbm.get("GRBM_GUI_ACTIVE", 0.0) / 8 / (gms * 1e-3) / 1e9 if gms else None,
    "wave_time": ({"parked_at_waitcnt_or_barrier": (sq.get("SQ_WAIT_ANY", 0) - sq.get("SQ_WAIT_INST_ANY", 0)) / waves if False else None,
                   "waiting_to_issue": sq.get("SQ_WAIT_INST_ANY", 0) / waves, "waiting_any": sq.get("SQ_WAIT_ANY", 0) / waves,
                   "issuing": sq.get("SQ_ACTIVE_INST_ANY", 0) / waves} if waves else None),
    "note": "counters of the LONGEST k_cdc_partial_grouped<double> dispatch of tools/bench_wide.py (the nrndm = 3000 launch) under rocprofv3 --pmc, one pass per "
            "counter group; HBM-side read bytes = 2 x FETCH_SIZE x 1024 on gfx950 (MI355X_MICROARCH.md)"}}
json.dump(rec, open(f"{R}/profiles/{T}_cdc_wide_counters.json", "w"), indent=1)
print(json.dumps(rec)[:600])
PY
cp $R/profiles/${T}_cdc_wide_counters.json $R/gpurun_out/ 2>/dev/null
