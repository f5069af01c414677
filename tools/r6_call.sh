cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "full" 2>&1 | tail -15
ONLY_LINEAR=1 DTYPE=f64 timeout 600 python tools/bench_full.py 2>&1 | tail -8
ONLY_LINEAR=1 DTYPE=f32 timeout 600 python tools/bench_full.py 2>&1 | tail -8
