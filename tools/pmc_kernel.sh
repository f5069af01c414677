#!/bin/bash
# on the GPU box: SQ / LDS counters of the kernels matching <pattern> while running <command>
# usage: tools/pmc_kernel.sh <kernel-name pattern[,pattern...]> '<command>'   (one block of lines per pattern)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
P=$1; CMD=$2
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pk_a /tmp/pk_b
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    --kernel-trace --output-format csv -d /tmp/pk_a -- bash -c "cd $R && $CMD" > /tmp/pk_a.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS \
    --kernel-trace --output-format csv -d /tmp/pk_b -- bash -c "cd $R && $CMD" > /tmp/pk_b.log 2>&1
python - "$P" <<'PY'
import csv, glob, sys, collections
rows = []
for d in ("a", "b"):
    for f in glob.glob(f"/tmp/pk_{d}/**/*counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
for pat in sys.argv[1].split(","):
    tot = collections.defaultdict(float); n = collections.defaultdict(int); dur = collections.defaultdict(float); names = set()
    for r in rows:
        if pat in r["Kernel_Name"]:
            names.add(r["Kernel_Name"][:110])
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
            dur[r["Counter_Name"]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    print(f"---- kernels matching '{pat}': {sorted(names)}")
    for k in sorted(tot):
        print(f"{k:24s} per dispatch {tot[k] / n[k]:14.4g}   dispatches {n[k]:4d}   avg ms {dur[k] / n[k]:8.3f}")
PY
tail -2 /tmp/pk_a.log
