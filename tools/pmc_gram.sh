#!/bin/bash
# on the GPU box: matrix-core counters of k_gram while tools/bench_gram.py runs one shape (ONLY=gram): MFMA busy cycles against the
# launch's clock cycles, MFMA instruction count, LDS bank conflicts.  usage: tools/pmc_gram.sh <tag> [SHAPES]   -> gpurun_out/<tag>_gram_pmc.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=$1; SH=${2:-50000,3000}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pg_a /tmp/pg_b /tmp/pg_c
CMD="cd $R && ONLY=gram REPS=2 SHAPES=$SH python tools/bench_gram.py"
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d /tmp/pg_a -- bash -c "$CMD" > /tmp/pg_a.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d /tmp/pg_b -- bash -c "$CMD" > /tmp/pg_b.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace --output-format csv -d /tmp/pg_c -- bash -c "$CMD" > /tmp/pg_c.log 2>&1
python - > $R/gpurun_out/${T}_gram_pmc.txt <<'PY'
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int); dur = collections.defaultdict(float)
for d in ("a", "b", "c"):
    for f in glob.glob(f"/tmp/pg_{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_gram<" in r["Kernel_Name"] or "k_gram_dma<" in r["Kernel_Name"]:
                key = (r["Kernel_Name"].split("(")[0], r["Counter_Name"])
                tot[key] += float(r["Counter_Value"]); n[key] += 1
                dur[key] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print("# rocprofv3 --pmc passes of: ONLY=gram REPS=2 python tools/bench_gram.py (per dispatch averages; k_gram_dma, or k_gram with VCY_GRAM_DMA=0)")
for (k, c) in sorted(tot):
    print(f"{k:70s} {c:28s} per dispatch {tot[(k, c)] / n[(k, c)]:14.6g}   dispatches {n[(k, c)]:3d}   avg ms {dur[(k, c)] / n[(k, c)]:9.3f}")
kern = sorted({k for k, _ in tot})
for k in kern:
    g = lambda c: (tot.get((k, c), 0.0) / max(1, n.get((k, c), 0)), dur.get((k, c), 0.0) / max(1, n.get((k, c), 0)))
    busy, ms = g("SQ_VALU_MFMA_BUSY_CYCLES")
    gui, ms2 = g("GRBM_GUI_ACTIVE")
    if busy and gui:
        clk = gui / 8.0 * (ms / ms2 if ms2 else 1.0)          # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs
        print(f"{k}: MFMA busy {busy:.4g} cycles / (1024 SIMDs x {clk:.4g} clocks of the launch) = {busy / (1024 * clk):.3f} of the matrix pipes' time; "
              f"effective clock {gui / 8.0 / (ms2 * 1e6):.3f} GHz")
PY
cat $R/gpurun_out/${T}_gram_pmc.txt | tail -12
tail -q -n 3 /tmp/pg_a.log /tmp/pg_b.log /tmp/pg_c.log | grep -i "error\|invalid\|fail" | head
