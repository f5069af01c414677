#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
tools/run_variants_cmd.sh 'python tools/bench_knn.py 2>&1 | grep "P= 30"'
python -m pytest -x -q -m gpu "tests/test_gpu_fullsize.py::test_fullsize_headline_arithmetic_against_the_oracle" tests/test_gpu_facade.py -k "hdf5 or serial or headline or checkpoint" 2>&1 | tail -3
python bench.py --dtype f32 --steps 3 --warmup 1 --cpu-cells 256 2>/dev/null | tail -1 | python -c "
import sys, json; j = json.loads(sys.stdin.read()); print(j['dtype'], j['value'], list(j['precision_modes'].keys()), {k: ('error' in v or 'skipped' in v) if isinstance(v, dict) else v for k, v in j['extra'].items()})"
