#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
SHORT=1 python tools/bench_pool.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_pool_f32.txt; cat gpurun_out/r04_pool_f32.txt
DTYPE=f64 python tools/bench_pool.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_pool_f64.txt; cat gpurun_out/r04_pool_f64.txt
for t in 8 16 32 64; do VCY_CHOICE_THREADS=$t python tools/bench_choice.py 2>&1 | grep -v amdgpu.ids | tail -1; done
