#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cat > /tmp/pool_one.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, velocyto_amd
from velocyto_amd import ops
import bench
dev = ops.require_gpu()
DT = torch.float64 if os.environ.get("DTYPE", "f32") == "f64" else torch.float32
C, G, k = 50000, 30000, 30
cS, cU, fS, fU, pcs = bench.synth_counts(C, G, 30, dev)
space = pcs[:, :30].contiguous()
idx, dist_ = ops.knn_search(space, k)
wrow = torch.cat([torch.ones((C, 1), device=dev), (dist_ > 0).float()], 1).to(DT)
wrow = (wrow / wrow.sum(1, keepdim=True)).contiguous()
indices = torch.cat([torch.arange(C, device=dev, dtype=torch.int32)[:, None], idx], 1).contiguous()
indptr = torch.arange(0, (C + 1) * (k + 1), k + 1, device=dev, dtype=torch.int64)
indices, wrow = ops.canonical_graph_rows(indices, wrow)
order = ops.hilbert_order(space)
o1, o2 = ops.CellMatrix.empty(C, G, DT), ops.CellMatrix.empty(C, G, DT)
for _ in range(3):
    ops.knn_pool_counts(cS, cU, fS, fU, indptr, indices, wrow, dtype=DT, out=o1, out2=o2, validate=False, order=order)
torch.cuda.synchronize()
PY
for D in f32 f64; do echo "== $D"; tools/pmc_kernel.sh k_knn_pool_counts "DTYPE=$D python /tmp/pool_one.py"; done > gpurun_out/r04_pool_pmc.txt 2>&1
cat gpurun_out/r04_pool_pmc.txt
