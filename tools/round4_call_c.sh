#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
STEPS=6 bash tools/run_variants.sh 2>&1 | grep -v amdgpu.ids > gpurun_out/r04b_elem_variants3.txt
cat gpurun_out/r04b_elem_variants3.txt
cp velocyto.py_amd/libvelocyto_hip.so /tmp/prod.so; cp velocyto.py_amd/libvelocyto_hip.exp6.so velocyto.py_amd/libvelocyto_hip.so
{ time timeout 1500 python -m pytest -q -x -m gpu tests/test_gpu_ops.py -k "cdc or coldeltacor or sqrt or partial or grouped" 2>&1 | tail -5 ; } 2>&1 | tail -9
cp /tmp/prod.so velocyto.py_amd/libvelocyto_hip.so
