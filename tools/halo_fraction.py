"""How many remote rows of e does a rank need when cells are sharded along a space-filling curve of the embedding? (default bench workload)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import velocyto_amd
from velocyto_amd import ops, distributed
import bench
dev = ops.require_gpu()
C = 50000
_, _, _, _, pcs = bench.synth_counts(C, 2000, 30, dev)
for curve, perm in (("morton", ops.morton_order(pcs[:, :2].contiguous(), 2).long()), ("hilbert", ops.hilbert_order(pcs[:, :2].contiguous()).long())):
  emb = pcs[perm][:, :2].contiguous()
  neigh, _ = bench.sample_neighbors_device(emb, 500, 0.5, dev)
  print(curve)
  for N in (2, 4, 8):
      fr = []
      for r in range(N):
          c0, c1 = distributed.shard_bounds(C, N, r)
          need = torch.zeros(C, dtype=torch.bool, device=dev)
          need[neigh[c0:c1].reshape(-1).long()] = True
          need[c0:c1] = False
          fr.append(int(need.sum()))
      print(f"N={N}: remote rows needed per rank: min {min(fr)} max {max(fr)} mean {sum(fr)/N:.0f}  "
            f"({sum(fr)/N/(C-C/N)*100:.1f} % of what an all-gather receives)")
