#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
{ time timeout 3000 python -m pytest -q -m gpu tests/ ; } > gpurun_out/r04d_tests_all.log 2>&1
tail -15 gpurun_out/r04d_tests_all.log
SHAPES="50000,3000;50000,10000;50000,30000" python tools/bench_gram.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_gram.txt; cat gpurun_out/r04_gram.txt
tools/pmc_gram.sh r04 50000,3000 > /dev/null 2>&1; tail -4 gpurun_out/r04_gram_pmc.txt
python tools/shard_model.py > gpurun_out/r04_shard_model.json 2> gpurun_out/r04_shard_model.err; python -c "
import json; d=json.load(open('gpurun_out/r04_shard_model.json'))
for n,w in d['worlds'].items(): print(n, round(w['predicted_ms_per_step'],1), round(w['predicted_cells_per_s']), round(w['speedup_vs_1'],2), round(w['efficiency'],3))"
