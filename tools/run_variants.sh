#!/bin/bash
# on the GPU box: time the default bench for the production library and for every libvelocyto_hip.exp*.so variant
# (EXTRA=1: with the f64 / uint16 / dual lines)
cd "$(dirname "$0")/.."
L=velocyto.py_amd/libvelocyto_hip.so
cp $L /tmp/prod.so
X="--no-extra"; [ -n "$EXTRA" ] && X=""
one() { python bench.py --no-cpu-baseline $X --steps ${STEPS:-3} --warmup 1 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); x=d.get('extra',{})
print('$1', round(d['value']), {k: round(v, 2) for k, v in d['config']['stage_ms'].items()}, 'dual', x.get('randomised_control',{}).get('D_dual_ms'), 'f64', x.get('f64',{}).get('ms_per_step'))"; }
one prod
for v in velocyto.py_amd/libvelocyto_hip.exp*.so; do [ -e $v ] || continue; cp $v $L; one $(basename $v); done
cp /tmp/prod.so $L
