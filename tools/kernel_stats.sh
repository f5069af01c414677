#!/bin/bash
# on the GPU box: rocprofv3 --kernel-trace --stats of a command, the kernels by total time
# usage: tools/kernel_stats.sh '<command>' [rows]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/ks
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- bash -c "cd $R && $1" > /tmp/ks.log 2>&1
f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1)
python - "$f" "${2:-25}" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:int(sys.argv[2])]:
    print("%9.2f ms  calls %5s  avg %8.3f ms  %s" % (float(r["TotalDurationNs"]) / 1e6, r["Calls"], float(r["AverageNs"]) / 1e6, r["Name"][:120]))
PY
grep -v amdgpu /tmp/ks.log | tail -16
