#!/bin/bash
# build an experimental variant of the library: tools/build_variant.sh <N> [extra flags] -> velocyto.py_amd/libvelocyto_hip.exp<N>.so
set -e
cd "$(dirname "$0")/.."
N=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -Iinclude -DVCY_EXP=$N "$@" -c velocyto.py_amd/csrc/coldeltacor.hip -o /tmp/cdc_exp$N.o
objs=$(ls velocyto.py_amd/csrc/_obj/*.o | grep -v coldeltacor.o)
hipcc --offload-arch=gfx950 -shared -fPIC /tmp/cdc_exp$N.o $objs -o velocyto.py_amd/libvelocyto_hip.exp$N.so
echo built exp$N
