#!/bin/bash
# On the GPU box: the rocprofv3 passes behind profiles/<tag>_*.csv (kernel stats, then one --pmc pass per counter group;
# counters are never combined with API tracing).  Usage: tools/profile_round.sh <tag>   -> gpurun_out/<tag>_{stats,fetch,write,sq}
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=$1
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-extra"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats -- $B > $R/gpurun_out/${T}_stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${T}_fetch -- $B --steps 1 --warmup 0 > $R/gpurun_out/${T}_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${T}_write -- $B --steps 1 --warmup 0 > $R/gpurun_out/${T}_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU \
    --kernel-trace --output-format csv -d $R/gpurun_out/${T}_sq -- $B --steps 1 --warmup 0 > $R/gpurun_out/${T}_sq.log 2>&1
cd $R
python bench.py 2>&1 | tail -1 > gpurun_out/${T}_bench_line.json
cut -c1-300 gpurun_out/${T}_bench_line.json
