#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
{ time timeout 2400 python -m pytest -x -q -m gpu "tests/test_gpu_ops.py::test_gene_quantiles_every_register_variant" "tests/test_gpu_ops.py::test_gene_quantiles" \
    "tests/test_gpu_ops.py::test_coldeltacor_partial_dual_equals_two_launches" "tests/test_gpu_ops.py::test_coldeltacor_partial_fused_dual" \
    "tests/test_gpu_fullsize.py::test_fullsize_stage_d_reference_default_list_width_against_the_oracle" ; } > gpurun_out/r04c_tests.log 2>&1
tail -8 gpurun_out/r04c_tests.log
{ python tools/bench_dual.py; LITERAL=1 python tools/bench_dual.py; DTYPE=f64 python tools/bench_dual.py; DTYPE=f64 VCY_CDC_DUAL_F64=0 python tools/bench_dual.py; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_stage_d_dual.txt
cat gpurun_out/r04_stage_d_dual.txt
for s in 512 2048; do python bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 --slab $s 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('slab', $s, 'f64 A_pooling_ms', j['config']['A_pooling_ms'], 'knn', j['config']['A_knn_search_ms'])"; done
DTYPE=f64 PASSES=2 C=50000 G=30000 PRE=0 python tools/run_facade.py 2>&1 | grep -v amdgpu.ids | tail -16
