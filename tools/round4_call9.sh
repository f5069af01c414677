#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
{ time timeout 1500 python -m pytest -x -q -m gpu "tests/test_gpu_ops.py::test_embedding_scaling_against_the_two_step_route" tests/test_gpu_facade.py ; } > gpurun_out/r04e_tests.log 2>&1
tail -6 gpurun_out/r04e_tests.log
{ python tools/bench_scaling.py; DTYPE=f64 python tools/bench_scaling.py; SINGLE=1 python tools/bench_scaling.py; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_scaling.txt; cat gpurun_out/r04_scaling.txt
tools/pmc_kernel.sh k_embedding_scaling "ONLY_NEW=1 python tools/bench_scaling.py" > gpurun_out/r04_scaling_pmc.txt 2>&1; cat gpurun_out/r04_scaling_pmc.txt | head -20
