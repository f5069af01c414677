#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
{ python tools/bench_dual.py; LITERAL=1 python tools/bench_dual.py; DTYPE=f64 python tools/bench_dual.py; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r04c_stage_d_dual.txt
cat gpurun_out/r04c_stage_d_dual.txt
python bench.py --steps 6 --warmup 2 --no-extra --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], {k: round(v, 2) for k, v in d['config']['stage_ms'].items()})"
{ time timeout 1500 python -m pytest -q -x -m gpu tests/test_gpu_ops.py tests/test_gpu_fullsize.py 2>&1 | tail -3 ; } 2>&1 | tail -7
