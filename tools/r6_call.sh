cd $GRAFT_REPO_ROOT
one() { python bench.py --no-cpu-baseline --no-extra --steps 4 --warmup 1 "${@:2}" 2>&1 | tail -1 > /tmp/line.json; python -c "
import sys,json; d=json.loads(open('/tmp/line.json').read())
print('$1', round(d['value']), {k: round(v, 2) for k, v in d['config']['stage_ms'].items()}, d['config']['count_layer_dtype'])
print('   telemetry', json.dumps(d.get('telemetry'))[:900])
print('   clock', json.dumps(d['roofline'].get('effective_clock'))[:500])" || tail -5 /tmp/line.json; }
one "generator bench"
one "generator survey" --generator survey
python - <<'PY'
import torch, bench, numpy as np
from velocyto_amd import ops
dev = torch.device("cuda", 0)
for name, fn in (("bench", lambda: bench.synth_counts(50000, 30000, 30, dev)), ("survey", lambda: bench.synth_counts_survey(50000, 30000, 30, dev))):
    cS, cU, fS, fU, pcs = fn()
    s = cS.as_int32(0, 8192)[:, :30000]; u = cU.as_int32(0, 8192)[:, :30000]
    print(name, "layer dtype", cS.t.dtype, "zeros S %.3f U %.3f" % (float((s == 0).float().mean()), float((u == 0).float().mean())), "max S", int(s.max()), "max U", int(u.max()),
          "mean S %.3f U %.3f" % (float(s.float().mean()), float(u.float().mean())), "pcs var ratio first/30th %.1f" % float(pcs[:, 0].var() / pcs[:, 29].var()))
PY
bash tools/pmc_wide.sh r06 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "default_list_width" 2>&1 | tail -3
