#!/bin/bash
# the round's closing call: whole GPU suite + smoke(), the profile passes (tools/profile_round.sh r06, tools/pmc_wide.sh r06), the driver-flag bench
# line and the side measurements (tools/measure_round.sh r06, the all-pairs kernels, the PCA products)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
T=r06
{ time timeout 2400 python -m pytest -q -m gpu ; } > gpurun_out/${T}_tests_all.log 2>&1          # from the root: pytest.ini keeps it to tests/
tail -6 gpurun_out/${T}_tests_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 > gpurun_out/${T}_smoke.txt; cat gpurun_out/${T}_smoke.txt
tools/profile_round.sh $T "round 6: f64 headline on uint16 count layers, SURVEY 8(d) generator (literal rule), f32 production mode" > gpurun_out/${T}_profile.log 2>&1; tail -2 gpurun_out/${T}_profile.log | cut -c1-300
bash tools/pmc_wide.sh $T > gpurun_out/${T}_pmc_wide.log 2>&1; tail -1 gpurun_out/${T}_pmc_wide.log | cut -c1-300
{ time python bench.py --gpus 1 --steps 20 --warmup 5 ; } > gpurun_out/${T}_bench.log 2> gpurun_out/${T}_bench.err; tail -1 gpurun_out/${T}_bench.log > gpurun_out/${T}_bench_line.json; tail -4 gpurun_out/${T}_bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 --generator bench --no-extra --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${T}_bench_line_generator_bench.json
tools/measure_round.sh $T > gpurun_out/${T}_measure.log 2>&1; tail -40 gpurun_out/${T}_measure.log | cut -c1-250
{ python tools/bench_full.py; DTYPE=f64 python tools/bench_full.py; } 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_full_kernels.txt; cat gpurun_out/${T}_full_kernels.txt
python tools/bench_pca_pass.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_pca_pass.txt; cat gpurun_out/${T}_pca_pass.txt
