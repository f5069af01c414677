#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
tools/run_variants_cmd.sh 'python -m pytest -q -m gpu tests/test_gpu_ops.py tests/test_gpu_atlas.py -k "pool" -x 2>&1 | tail -2; SHORT=1 python tools/bench_pool.py | grep -E "slab  1024|slab  2048|one neighbour|all-self"; DTYPE=f64 python tools/bench_pool.py | grep -E "slab  1024|slab  2048|one neighbour|all-self"' > gpurun_out/r04_pool_variants.txt 2>&1
cat gpurun_out/r04_pool_variants.txt
