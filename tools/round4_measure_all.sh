#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
tools/profile_round.sh r04 "round 4: f64 headline (literal rule, square root from the f32 product seed, zero rule on the seed argument, four-lane LDS adds, one partial sum per moment), f32 production mode" > gpurun_out/r04_profile.log 2>&1; tail -2 gpurun_out/r04_profile.log | cut -c1-300
{ time python bench.py --gpus 1 --steps 20 --warmup 5 ; } > gpurun_out/r04_bench.log 2> gpurun_out/r04_bench.err; tail -1 gpurun_out/r04_bench.log > gpurun_out/r04_bench_line.json; tail -4 gpurun_out/r04_bench.err
tools/measure_round.sh r04 > gpurun_out/r04_measure.log 2>&1; tail -40 gpurun_out/r04_measure.log | cut -c1-250
