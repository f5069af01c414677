#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
{ time timeout 1500 python -m pytest -x -q -m gpu "tests/test_gpu_ops.py::test_embedding_scaling_against_the_two_step_route" tests/test_gpu_facade.py ; } > gpurun_out/r04e_tests.log 2>&1
tail -12 gpurun_out/r04e_tests.log
PASSES=3 C=50000 G=30000 PRE=0 python tools/run_facade.py 2>&1 | grep -v amdgpu.ids | tail -16
DTYPE=f64 PASSES=2 C=50000 G=30000 PRE=0 python tools/run_facade.py 2>&1 | grep -v amdgpu.ids | tail -16
