#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for f in "" "--no-fuse"; do
python bench.py --steps 6 --warmup 2 --no-extra --no-cpu-baseline $f 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$f', round(d['value']), d['ms_per_step'], {k: round(v, 2) for k, v in d['config']['stage_ms'].items()})"
done
/tmp/vi 2>/dev/null | tail -3
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/valu_issue_f64.hip -o /tmp/vi 2>/dev/null && /tmp/vi > gpurun_out/r04b_valu_issue_f64.txt 2>&1
tail -9 gpurun_out/r04b_valu_issue_f64.txt | cut -c1-330
