"""Distribution of the synthetic count layers (what a narrower resident encoding would have to cover)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import velocyto_amd
from velocyto_amd import ops
import bench
dev = ops.require_gpu()
cS, cU, fS, fU, pcs = bench.synth_counts(50000, 30000, 30, dev)
for name, c in (("S", cS), ("U", cU)):
    v = c.as_int32()
    n = v.numel()
    print(name, "zeros %.3f" % ((v == 0).sum().item() / n), " <16 %.4f" % ((v < 16).sum().item() / n), " <256 %.6f" % ((v < 256).sum().item() / n),
          " max", int(v.max()), " mean %.3f" % v.float().mean().item())
