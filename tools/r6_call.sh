cd $GRAFT_REPO_ROOT
L=velocyto.py_amd/libvelocyto_hip.so
cp $L /tmp/prod.so
one() { python bench.py --no-cpu-baseline --no-extra --steps 4 --warmup 1 --dump /tmp/dump_$2.npz 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('$1', round(d['value']), 'A_pooling %.2f  A_knn_search %.2f  B %.2f  D %.2f' % (d['config']['A_pooling_ms'], d['config']['A_knn_search_ms'], d['config']['B_fit_slope_ms'], d['config']['D_coldeltacor_ms']))"; }
one "prod                         " a
one "prod again                   " a2
cp velocyto.py_amd/libvelocyto_hip.exp42.so $L
one "probe: U launch reads Sx + 24 FMA " b
one "again                        " b2
cp /tmp/prod.so $L
python -c "
import numpy as np
a, b = np.load('/tmp/dump_a.npz'), np.load('/tmp/dump_b.npz')
print('gamma identical:', np.array_equal(a['gamma'], b['gamma']), ' corr identical:', np.array_equal(a['corr'], b['corr'], equal_nan=True))"
