#!/bin/bash
# On the GPU box: the rocprofv3 passes behind profiles/<tag>_*.csv (kernel stats, then one --pmc pass per counter group;
# counters are never combined with API tracing).  f32 (the production mode) and f64 (the reference's arithmetic) each get the
# full set, so that both stage-D kernels have a kernel_stats row, counters and an effective clock.
# Usage: tools/profile_round.sh <tag>   -> gpurun_out/<tag>_{stats,fetch,write,sq,grbm}[_f64]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=$1
cd /tmp; export TMPDIR=/tmp
for D in f32 f64; do
  S=""; [ $D = f64 ] && S="_f64"
  B="python $R/bench.py --no-cpu-baseline --no-extra --dtype $D"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats$S -- $B > $R/gpurun_out/${T}_stats$S.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${T}_fetch$S -- $B --steps 1 --warmup 0 > $R/gpurun_out/${T}_fetch$S.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${T}_write$S -- $B --steps 1 --warmup 0 > $R/gpurun_out/${T}_write$S.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU \
      --kernel-trace --output-format csv -d $R/gpurun_out/${T}_sq$S -- $B --steps 1 --warmup 0 > $R/gpurun_out/${T}_sq$S.log 2>&1
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/${T}_grbm$S -- $B --steps 1 --warmup 0 > $R/gpurun_out/${T}_grbm$S.log 2>&1
done
# only the CSVs travel back (gpurun merges at most 64 MiB)
find $R/gpurun_out -name "*.db" -delete 2>/dev/null
cd $R
python tools/summarize_profiles.py $T "${2:-}" > gpurun_out/${T}_summarize.log 2>&1     # writes profiles/<tag>_cdc_counters.json on the box, so that
cp profiles/${T}_* gpurun_out/ 2>/dev/null                                                # the bench line below is computed from THIS build's counters
python bench.py 2>&1 | tail -1 > gpurun_out/${T}_bench_line.json
cut -c1-300 gpurun_out/${T}_bench_line.json
