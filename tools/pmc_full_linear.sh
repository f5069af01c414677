#!/bin/bash
# on the GPU box: matrix-core / LDS / wait counters of k_cdc_full_linear while tools/bench_full.py runs the linear variant (f64, 10 000 x 20 000)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pf_a /tmp/pf_b
CMD="cd $R && ONLY_LINEAR=1 DTYPE=f64 python tools/bench_full.py"
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d /tmp/pf_a -- bash -c "$CMD" > /tmp/pf_a.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/pf_b -- bash -c "$CMD" > /tmp/pf_b.log 2>&1
python - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int); dur = collections.defaultdict(float)
for d in ("a", "b"):
    for f in glob.glob(f"/tmp/pf_{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_cdc_full_linear" in r["Kernel_Name"]:
                c = r["Counter_Name"]
                tot[c] += float(r["Counter_Value"]); n[c] += 1; dur[c] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for c in sorted(tot):
    print(f"{c:28s} per dispatch {tot[c] / n[c]:14.6g}   dispatches {n[c]:3d}   avg ms {dur[c] / n[c]:9.3f}")
if "SQ_VALU_MFMA_BUSY_CYCLES" in tot and "GRBM_GUI_ACTIVE" in tot:
    busy, ms = tot["SQ_VALU_MFMA_BUSY_CYCLES"] / n["SQ_VALU_MFMA_BUSY_CYCLES"], dur["SQ_VALU_MFMA_BUSY_CYCLES"] / n["SQ_VALU_MFMA_BUSY_CYCLES"]
    gui, ms2 = tot["GRBM_GUI_ACTIVE"] / n["GRBM_GUI_ACTIVE"], dur["GRBM_GUI_ACTIVE"] / n["GRBM_GUI_ACTIVE"]
    clk = gui / 8.0 * ms / ms2
    print(f"MFMA busy {busy:.4g} cycles / (1024 SIMDs x {clk:.4g} clocks of the launch) = {busy / (1024 * clk):.3f} of the matrix pipes' time")
PY
