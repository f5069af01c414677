#!/bin/bash
# on the GPU box: run a command once for the production library and once for every libvelocyto_hip.exp*.so variant
# usage: tools/run_variants_cmd.sh '<command printing what you want to compare>'
cd "$(dirname "$0")/.."
L=velocyto.py_amd/libvelocyto_hip.so
cp $L /tmp/prod.so
echo "== prod"; bash -c "$1" 2>&1 | grep -v amdgpu.ids
for v in velocyto.py_amd/libvelocyto_hip.exp*.so; do [ -e $v ] || continue; cp $v $L; echo "== $(basename $v)"; bash -c "$1" 2>&1 | grep -v amdgpu.ids; done
cp /tmp/prod.so $L
