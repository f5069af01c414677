#!/bin/bash
# build an experimental variant of the library: tools/build_variant.sh <N> [extra flags] -> velocyto.py_amd/libvelocyto_hip.exp<N>.so
# (SRC=<file stem under csrc/> picks the source compiled with -DVCY_EXP=<N> and the extra flags; default coldeltacor)
set -e
cd "$(dirname "$0")/.."
N=$1; shift
S=${SRC:-coldeltacor}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -Iinclude -DVCY_EXP=$N "$@" -c velocyto.py_amd/csrc/$S.hip -o /tmp/${S}_exp$N.o
objs=$(ls velocyto.py_amd/csrc/_obj/*.o | grep -v "/$S.o")
hipcc --offload-arch=gfx950 -shared -fPIC /tmp/${S}_exp$N.o $objs -o velocyto.py_amd/libvelocyto_hip.exp$N.so
echo built exp$N
