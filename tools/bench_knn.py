"""Micro-benchmark of k_knn_search: feature dimension P and k sweep (phase 1 ~ P, phase 2 ~ independent of P)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import velocyto_amd
from velocyto_amd import ops
dev = ops.require_gpu()
C = int(os.environ.get("C", 50000))
g = torch.Generator(device=dev).manual_seed(0)
for P in (2, 8, 30, 60):
    X = torch.randn((C, P), generator=g, device=dev, dtype=torch.float64)
    for k in (30, 501):
        ts = []
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ops.knn_search(X, k)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"P={P:3d} k={k:4d}  {min(ts)*1e3:8.2f} ms  ({C} queries)")
