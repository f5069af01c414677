#!/bin/bash
# the round's closing call: whole GPU suite + smoke(), then the profile passes (tools/profile_round.sh r05), the driver-flag bench line and the
# side measurements (tools/measure_round.sh r05 + the all-pairs kernels)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
{ time timeout 2400 python -m pytest -q -m gpu tests/ ; } > gpurun_out/r05_tests_all.log 2>&1
tail -6 gpurun_out/r05_tests_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 > gpurun_out/r05_smoke.txt; cat gpurun_out/r05_smoke.txt
tools/profile_round.sh r05 "round 5: f64 headline on uint16 count layers (literal rule; the last rows of a chunk drawn in pairs), f32 production mode" > gpurun_out/r05_profile.log 2>&1; tail -2 gpurun_out/r05_profile.log | cut -c1-300
{ time python bench.py --gpus 1 --steps 20 --warmup 5 ; } > gpurun_out/r05_bench.log 2> gpurun_out/r05_bench.err; tail -1 gpurun_out/r05_bench.log > gpurun_out/r05_bench_line.json; tail -4 gpurun_out/r05_bench.err
tools/measure_round.sh r05 > gpurun_out/r05_measure.log 2>&1; tail -40 gpurun_out/r05_measure.log | cut -c1-250
{ python tools/bench_full.py; DTYPE=f64 python tools/bench_full.py; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_full_kernels.txt; cat gpurun_out/r05_full_kernels.txt
