#!/bin/bash
# GPU call 1 of round 4: the new tests, the new bench line (driver's flags), the Gram kernel alone, then the rocprofv3 passes of the round
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
{ time timeout 1500 python -m pytest -x -q -m gpu tests/test_gpu_preprocess.py tests/test_gpu_bench_line.py "tests/test_gpu_ops.py::test_f64_sqrt_element_accuracy_and_domain" \
    "tests/test_gpu_atlas.py::test_knn_pool_csr_layers_with_a_handful_of_nonzeros" "tests/test_gpu_atlas.py::test_csr_counts_container" \
    "tests/test_gpu_facade.py::test_facade_fit_gammas_steady_state" "tests/test_gpu_distributed.py::test_failed_collective_self_check_falls_back_to_allgather" \
    "tests/test_gpu_distributed.py::test_rccl_collectives_on_one_gpu" \
    "tests/test_gpu_fullsize.py::test_fullsize_stage_d_reference_default_list_width_against_the_oracle" ; } > gpurun_out/r04a_tests.log 2>&1
tail -15 gpurun_out/r04a_tests.log
{ time python bench.py --gpus 1 --steps 20 --warmup 5 ; } > gpurun_out/r04a_bench.log 2> gpurun_out/r04a_bench.err
tail -1 gpurun_out/r04a_bench.log | cut -c1-600; tail -4 gpurun_out/r04a_bench.err
tail -1 gpurun_out/r04a_bench.log > gpurun_out/r04a_bench_line.json
python tools/bench_gram.py > gpurun_out/r04a_gram.txt 2>&1; grep -v amdgpu.ids gpurun_out/r04a_gram.txt
tools/pmc_gram.sh r04a 50000,3000
tools/profile_round.sh r04 "round 4, first pass" > gpurun_out/r04a_profile.log 2>&1; tail -3 gpurun_out/r04a_profile.log
