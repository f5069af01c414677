#!/bin/bash
# On the GPU box: everything the round's documents quote besides the rocprofv3 passes (tools/profile_round.sh <tag> runs first):
# the 8-rank rehearsal on one device, the cfg5 lines, stage D on shard shapes / wide lists / with the randomised control, the shard model,
# the facade timings.
# Usage: tools/measure_round.sh <tag>   -> gpurun_out/<tag>_*
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=$1
cd $R
VCY_SINGLE_DEVICE=1 VCY_DIST_BACKEND=gloo python bench.py --gpus 8 --no-cpu-baseline --no-extra --steps 2 --warmup 1 2> gpurun_out/${T}_8ranks.err | tail -1 > gpurun_out/${T}_bench_8ranks_one_device_line.json
# cfg5 in the build's production arithmetic (f32: the round-2/3 lines) and, for the full 1M cells, in the reference's (f64, bench.py's default)
python bench.py --workload cfg5 --dtype f32 --cells 200000 --no-cpu-baseline --steps 2 --warmup 1 2> gpurun_out/${T}_cfg5_200k.err | tail -1 > gpurun_out/${T}_bench_cfg5_200k_line.json
python bench.py --workload cfg5 --dtype f32 --cells 1000000 --no-cpu-baseline --steps 2 --warmup 1 2> gpurun_out/${T}_cfg5_1M.err | tail -1 > gpurun_out/${T}_bench_cfg5_1M_line.json
python bench.py --workload cfg5 --dtype f64 --cells 1000000 --no-cpu-baseline --steps 1 --warmup 1 2> gpurun_out/${T}_cfg5_1M_f64.err | tail -1 > gpurun_out/${T}_bench_cfg5_1M_f64_line.json
{ python tools/bench_shapes.py; DTYPE=f64 python tools/bench_shapes.py; } 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_stage_d_shapes.txt
{ python tools/bench_dual.py; LITERAL=1 python tools/bench_dual.py; DTYPE=f64 python tools/bench_dual.py; } 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_stage_d_dual.txt
{ python tools/bench_scaling.py; DTYPE=f64 python tools/bench_scaling.py; } 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_stage_e_scaling.txt
python tools/bench_gram.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_gram.txt
python tools/shard_model.py > gpurun_out/${T}_shard_model.json 2> gpurun_out/${T}_shard_model.err
# the facade around the path: three passes (first calls with their allocations, then steady state), the per-op breakdown, the sampling
# replay alone, and the kernels of one pass by total time
{ echo "# MEM=1 PASSES=3 C=50000 G=30000 PRE=0 python tools/run_facade.py   (count layers in, f32 storage)"; MEM=1 PASSES=3 C=50000 G=30000 PRE=0 python tools/run_facade.py 2>&1 | grep -v amdgpu.ids
  echo; echo "# DTYPE=f64 PASSES=2 C=50000 G=30000 PRE=0 python tools/run_facade.py   (f64 storage: the reference's arithmetic)"; DTYPE=f64 PASSES=2 C=50000 G=30000 PRE=0 python tools/run_facade.py 2>&1 | grep -v amdgpu.ids
  echo; echo "# python tools/facade_breakdown.py  (steady state, every ops call device-synchronised)"; python tools/facade_breakdown.py 2>&1 | grep -v amdgpu.ids | awk '/^normalize/{n++} n>=2'
  echo; echo "# python tools/bench_choice.py; VCY_CHOICE_THREADS=1 python tools/bench_choice.py"; python tools/bench_choice.py 2>&1 | grep -v amdgpu.ids | tail -1; VCY_CHOICE_THREADS=1 python tools/bench_choice.py 2>&1 | grep -v amdgpu.ids | tail -1
} > gpurun_out/${T}_facade_50k.txt
tools/kernel_stats.sh "C=50000 G=30000 PRE=0 python tools/run_facade.py" 32 > gpurun_out/${T}_facade_kernel_stats.txt 2>&1
python tools/bench_markov.py > gpurun_out/${T}_markov_steps.txt 2>&1
for f in gpurun_out/${T}_bench_8ranks_one_device_line.json gpurun_out/${T}_bench_cfg5_200k_line.json gpurun_out/${T}_bench_cfg5_1M_line.json gpurun_out/${T}_bench_cfg5_1M_f64_line.json; do cut -c1-250 $f; echo; done
cat gpurun_out/${T}_stage_d_shapes.txt gpurun_out/${T}_stage_d_dual.txt | grep -v amdgpu.ids
head -48 gpurun_out/${T}_facade_50k.txt
