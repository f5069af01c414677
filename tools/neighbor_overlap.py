import sys, torch, numpy as np
sys.path.insert(0,''+__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))+'')
import velocyto_amd
from velocyto_amd import ops
import bench
dev=ops.require_gpu()
C,G=50000,3000
S,U,pcs=bench.synth(C,G,30,dev)
emb=pcs[:,:2].contiguous()
neigh,_=bench.sample_neighbors_device(emb,500,0.5,dev)
order=ops.morton_order(emb,2).long()
nb=neigh[order].cpu().numpy()
for GC in (2,4,8,16,32):
    n=(C//GC)*GC
    grp=nb[:n].reshape(-1,GC*nb.shape[1])
    distinct=np.array([len(np.unique(r)) for r in grp[::50]])
    print(GC, "pairs",GC*nb.shape[1],"distinct mean",distinct.mean(),"mult",GC*nb.shape[1]/distinct.mean())
