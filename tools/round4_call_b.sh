#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
bash tools/pmc_kernel.sh 'k_cdc_partial_grouped<double' 'python bench.py --steps 1 --warmup 0 --no-extra --no-cpu-baseline' > gpurun_out/r04b_cdc_f64_pmc.txt 2>&1
cat gpurun_out/r04b_cdc_f64_pmc.txt
