#!/bin/bash
# GPU call 2 of round 4: the tests call 1 did not reach, the facade after the lazy graph containers, the f64 ubench with the new square root
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
{ time timeout 2400 python -m pytest -x -q -m gpu "tests/test_gpu_ops.py::test_f64_sqrt_element_accuracy_and_domain" \
    "tests/test_gpu_atlas.py::test_knn_pool_csr_layers_with_a_handful_of_nonzeros" "tests/test_gpu_atlas.py::test_csr_counts_container" \
    tests/test_gpu_facade.py "tests/test_gpu_distributed.py::test_failed_collective_self_check_falls_back_to_allgather" \
    "tests/test_gpu_distributed.py::test_rccl_collectives_on_one_gpu" tests/test_loom_io.py \
    "tests/test_gpu_fullsize.py::test_fullsize_stage_d_reference_default_list_width_against_the_oracle" \
    "tests/test_gpu_fullsize.py::test_fullsize_stage_d_256_cells_against_the_oracle" ; } > gpurun_out/r04b_tests.log 2>&1
tail -12 gpurun_out/r04b_tests.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_issue_f64 tools/ubench/valu_issue_f64.hip 2>/dev/null && /tmp/valu_issue_f64 > gpurun_out/r04_valu_issue_f64.txt 2>&1; tail -7 gpurun_out/r04_valu_issue_f64.txt
{ echo "# MEM=1 PASSES=3 C=50000 G=30000 PRE=0 python tools/run_facade.py   (count layers in, f32 storage)"; MEM=1 PASSES=3 C=50000 G=30000 PRE=0 python tools/run_facade.py 2>&1 | grep -v amdgpu.ids
  echo; echo "# DTYPE=f64 PASSES=2 ..."; DTYPE=f64 PASSES=2 C=50000 G=30000 PRE=0 python tools/run_facade.py 2>&1 | grep -v amdgpu.ids
  echo; echo "# python tools/facade_breakdown.py  (steady state, every ops call device-synchronised)"; python tools/facade_breakdown.py 2>&1 | grep -v amdgpu.ids | awk '/^normalize/{n++} n>=2'
} > gpurun_out/r04b_facade_50k.txt 2>&1
grep -A16 "pass 3" gpurun_out/r04b_facade_50k.txt | head -20; grep -B2 -A8 "^knn_imputation:" gpurun_out/r04b_facade_50k.txt | tail -12
python bench.py --no-cpu-baseline --no-extra --steps 5 --warmup 2 2>/dev/null | tail -1 | cut -c1-400
