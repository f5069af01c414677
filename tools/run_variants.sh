#!/bin/bash
# on the GPU box: time stage D for every libvelocyto_hip.exp*.so variant (and the production library first)
cd "$(dirname "$0")/.."
L=velocyto.py_amd/libvelocyto_hip.so
cp $L /tmp/prod.so
one() { python bench.py --no-cpu-baseline --no-extra --steps ${STEPS:-3} --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), d['config']['stage_ms'])"; }
one prod
for v in velocyto.py_amd/libvelocyto_hip.exp*.so; do cp $v $L; one $(basename $v); done
cp /tmp/prod.so $L
