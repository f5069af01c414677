cd $GRAFT_REPO_ROOT
one() { python bench.py --no-cpu-baseline --no-extra --steps 4 --warmup 1 "${@:2}" 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('$1', round(d['value']), {k: round(v, 2) for k, v in d['config']['stage_ms'].items()})"; }
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "partial or dual or fused" 2>&1 | tail -3
one "LPT rows f64"
one "LPT rows f64 again"
one "LPT rows f32" --dtype f32
