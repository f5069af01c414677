"""Time the replay of numpy's neighbour-sampling stream (ops.choice_stream_host: estimate_transition_prob's host side) alone.
usage: [C=50000 N=501 SIZE=250] python tools/bench_choice.py     (VCY_CHOICE_THREADS=1: the sequential replay)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from velocyto_amd import ops

C, n, size = int(os.environ.get("C", 50000)), int(os.environ.get("N", 501)), int(os.environ.get("SIZE", 250))
p = np.linspace(0.5, 0.1, n)
p /= p.sum()
t = time.perf_counter(); np.random.seed(1); x = np.random.random_sample(C * 316); rng = time.perf_counter() - t
best = 1e9
for rep in range(5):
    np.random.seed(15071990)
    t = time.perf_counter(); c = time.process_time()
    out = ops.choice_stream_host(n, size, p, C)
    w, cpu = time.perf_counter() - t, time.process_time() - c
    best = min(best, w)
    print(f"{C} cells, {size} of {n}: wall {1e3 * w:.1f} ms, cpu {1e3 * cpu:.1f} ms (threads {os.environ.get('VCY_CHOICE_THREADS', '8')}); drawing the uniforms alone {1e3 * rng:.1f} ms; checksum {int(out.sum())}")
