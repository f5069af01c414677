import sys, time, torch, numpy as np
sys.path.insert(0, ".")
from velocyto_amd import ops
n = 50000
for dt in (torch.float32, torch.float64):
    T = torch.rand((n, n), dtype=dt, device="cuda")
    T /= T.sum(1, keepdim=True)
    x0 = np.ones(n) / n
    ops.diffuse(x0, T, 40, False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ops.diffuse(x0, T, 400, False)
    torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / 400
    print(dt, f"{dtm*1e3:.3f} ms/step  {n*n*T.element_size()/dtm/1e12:.2f} TB/s")
    del T
