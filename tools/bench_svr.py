"""Device SVR (vcy_svr_rbf_fit / _predict) against scikit-learn on the two shapes the reference fits:
score_cv_vs_mean (one point per gene, gamma = 150/G) and adjust_totS_totU (one point per cell, C = 100, gamma = 1e-6).
usage: python tools/bench_svr.py [n ...]   (VCY_SVR_WG=k forces the number of workgroups)"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from velocyto_amd import ops


def cases(n, rng):
    x = np.log2(rng.gamma(0.5, 0.5, n) + 1e-3); t = -0.5 * x + 0.3 * rng.normal(size=n) + 0.5 * np.exp(-x * x)
    yield "cv_vs_mean", x, t, dict(C=1.0, gamma=150.0 / n)
    x = rng.gamma(5, 2000, n); t = 0.3 * x * (1 + 0.2 * np.sin(x / 5000)) + rng.normal(0, 300, n)
    yield "totS_totU", x, t, dict(C=100.0, gamma=1e-6)


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [2000, 10000]
    with_sklearn = True
    rng = np.random.default_rng(0)
    for n in sizes:
        for name, x, t, kw in cases(n, rng):
            for rep in range(2):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                coef, b, info = ops.svr_fit(x, t, **kw)
                pred = ops.svr_predict(x, coef, b, x, kw["gamma"])
                torch.cuda.synchronize(); t1 = time.perf_counter()
            info = info.cpu().numpy(); pred = pred.cpu().numpy()
            line = f"{name:11s} n={n:6d}: device {t1 - t0:7.3f} s  steps {info[0]} converged {info[1]} failed {info[2]} wgs {info[3]}"
            if with_sklearn and n <= 20000:
                from sklearn.svm import SVR
                t2 = time.perf_counter(); sk = SVR(**kw).fit(x[:, None], t); ps = sk.predict(x[:, None]); t3 = time.perf_counter()
                line += f" | sklearn {t3 - t2:6.2f} s  max|dpred| {np.abs(pred - ps).max():.2e} (scale {np.abs(ps).max():.2e})  nSV {int((coef != 0).sum())} vs {len(sk.support_)}"
            print(line, flush=True)


if __name__ == "__main__":
    main()
