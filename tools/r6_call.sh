cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_preprocess.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python tools/bench_pca_pass.py 2>&1 | grep -v amdgpu.ids
python - <<'PY'
import time, torch, velocyto_amd
from velocyto_amd import ops
from velocyto_amd.preprocess import DevicePCA
dev = ops.require_gpu()
C, G = 50000, 30000
X = ops.CellMatrix.empty(C, G, torch.float64)
X.t[:, :G] = torch.rand((C, 8), device=dev, dtype=torch.float64) @ torch.rand((8, G), device=dev, dtype=torch.float64) + 0.1 * torch.rand((C, G), device=dev, dtype=torch.float64)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    p = DevicePCA(n_components=30); p.fit_transform(X); torch.cuda.synchronize()
    print("DevicePCA(n_components=30).fit_transform at 50000 x 30000 f64: %.1f ms, %d passes" % ((time.perf_counter() - t0) * 1e3, p.n_iter_))
PY
