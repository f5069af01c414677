#!/bin/bash
# sweep the pooling slab width (genes per pass) on the default bench workload
for sl in 64 128 256 512; do
  python bench.py --no-cpu-baseline --slab $sl 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('slab', $sl, round(d['value']), d['config']['stage_ms']['A_knn_imputation'])"
done
