#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/valu_issue_f64.hip -o /tmp/vi 2>/dev/null && /tmp/vi > gpurun_out/r04b_valu_issue_f64.txt 2>&1
tail -7 gpurun_out/r04b_valu_issue_f64.txt | cut -c1-400
{ time timeout 1500 python -m pytest -q -x -m gpu tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py 2>&1 | tail -5 ; } 2>&1 | tail -9
python bench.py --steps 10 --warmup 3 --no-extra 2>&1 | tail -1 > gpurun_out/r04b_line.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04b_line.json').read())
print(d['value'], d['ms_per_step'], {k:v for k,v in d['config'].items() if k.endswith('_ms')}, d['roofline'])
PY
