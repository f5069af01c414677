#!/bin/bash
# on the GPU box: SQ / LDS counters of the stage-D kernel for the production library and every libvelocyto_hip.exp*.so variant
cd "$(dirname "$0")/.."
R=$PWD
L=velocyto.py_amd/libvelocyto_hip.so
cp $L /tmp/prod.so
export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-extra --steps 1 --warmup 0"
one() {
  rm -rf /tmp/pm_$1_a /tmp/pm_$1_b
  (cd /tmp && timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
      --kernel-trace --output-format csv -d /tmp/pm_$1_a -- $B > /tmp/pm_$1_a.log 2>&1)
  (cd /tmp && timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS \
      --kernel-trace --output-format csv -d /tmp/pm_$1_b -- $B > /tmp/pm_$1_b.log 2>&1)
  python - "$1" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
tot = collections.defaultdict(float); dur = {}
for d in ("a", "b"):
    for f in glob.glob(f"/tmp/pm_{tag}_{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_cdc_partial_grouped" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
    for f in glob.glob(f"/tmp/pm_{tag}_{d}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_cdc_partial_grouped" in r["Kernel_Name"]:
                dur[d] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print(tag, "ms", dur, {k: f"{v:.4g}" for k, v in sorted(tot.items())})
PY
}
one prod
for v in velocyto.py_amd/libvelocyto_hip.exp*.so; do cp $v $L; one $(basename $v .so | sed 's/libvelocyto_hip.//'); done
cp /tmp/prod.so $L
tail -3 /tmp/pm_prod_b.log
