cd $GRAFT_REPO_ROOT
L=velocyto.py_amd/libvelocyto_hip.so
cp $L /tmp/prod.so
python tools/r6_markov.py table 2>&1 | grep -v amdgpu.ids
cp velocyto.py_amd/libvelocyto_hip.exp31.so $L
python tools/r6_markov.py poly13 2>&1 | grep -v amdgpu.ids
cp /tmp/prod.so $L
