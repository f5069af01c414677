"""Stage D alone on the shapes the documents quote besides the headline: the first 1/8, 1/4, 1/2 and all of the 50 000 cells
(what one rank of an 8 / 4 / 2 / 1-GPU run computes; last-round tiles) and the reference's default list width
(n_neighbors = C/5, sampled_fraction = 0.3 -> nrndm = 3000, walked in column tiles)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import velocyto_amd
from velocyto_amd import ops
import bench

dev = ops.require_gpu()
C, G = int(os.environ.get("C", 50000)), int(os.environ.get("G", 30000))
DT = torch.float64 if os.environ.get("DTYPE", "f32") == "f64" else torch.float32
S, U, pcs = bench.synth(C, G, 30, dev)
del U
if DT == torch.float64:
    S = ops.CellMatrix(S.t.double(), G)
emb = pcs[:, :2].contiguous()
d = ops.CellMatrix(torch.randn_like(S.t), G)
print("storage:", os.environ.get("DTYPE", "f32"))
rules = ops.partial_rules_for(S, ops.SQRT, 1e-10)
print("rule:", ops.RULE_NAMES[rules])


def best(f, reps=3):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3


def clock(f, ms):
    """Shader clock (mean GHz over 2 ms readings, vcy_clock_probe) while f runs."""
    pr = ops.ClockProbe()
    torch.cuda.synchronize()
    pr.start(0.9 * ms)
    f()
    torch.cuda.synchronize()
    return pr.ghz()[0]


neigh, _ = bench.sample_neighbors_device(emb, 500, 0.5, dev)
perm = ops.hilbert_order(emb).long()                     # a rank's cells are a contiguous piece of the curve
for n in (C // 8, C // 4, C // 2, C):
    cells = perm[:n].to(torch.int32).contiguous()        # schedule over a subset: only those rows are written
    out = torch.empty((C, neigh.shape[1]), dtype=DT, device=dev)
    ms = best(lambda: ops.coldeltacor_partial(S, d, neigh, ops.SQRT, rules, 1e-10, order=cells, out=out, validate=False))
    ghz = clock(lambda: ops.coldeltacor_partial(S, d, neigh, ops.SQRT, rules, 1e-10, order=cells, out=out, validate=False), ms)
    print(f"nrndm {neigh.shape[1]:5d}  cells {n:6d}  {ms:8.2f} ms  {ms / n * 1e3:6.3f} us per cell   shader clock {ghz:.2f} GHz")
if os.environ.get("POOLED"):
    # the same launch on POOLED matrices (what the pipeline hands stage D: every cell a weighted mean of 31 cells, hardly an exact zero left)
    import scipy.sparse  # noqa: F401
    idx, dist_ = ops.knn_search(pcs, 30, include_self=False)
    w = torch.cat([torch.ones((C, 1), device=dev, dtype=DT), (dist_ > 0).to(DT)], 1)
    w = w / w.sum(1, keepdim=True)
    ind = torch.cat([torch.arange(C, device=dev, dtype=torch.int32)[:, None], idx], 1)
    ind, w = ops.canonical_graph_rows(ind, w)
    ptr = torch.arange(0, (C + 1) * 31, 31, device=dev, dtype=torch.int64)
    Sp = ops.knn_pool(S, ptr, ind, w)
    out = torch.empty((C, neigh.shape[1]), dtype=DT, device=dev)
    order = ops.hilbert_order(emb)
    for name, E in (("unpooled e (S_sz: counts x size factor)", S), ("pooled e (Sx_sz)", Sp)):
        f = lambda: ops.coldeltacor_partial(E, d, neigh, ops.SQRT, rules, 1e-10, order=order, out=out, validate=False)
        ms = best(f)
        print(f"{name:42s} {ms:8.2f} ms   shader clock {clock(f, ms):.2f} GHz   exact zeros in e: {float((E.t[:, :G] == 0).double().mean()):.3f}")
if not os.environ.get("SKIP_WIDE"):
    wide, _ = bench.sample_neighbors_device(emb, C // 5, 0.3, dev)
    out = torch.empty((C, wide.shape[1]), dtype=DT, device=dev)
    order = ops.hilbert_order(emb)
    ms = best(lambda: ops.coldeltacor_partial(S, d, wide, ops.SQRT, rules, 1e-10, order=order, out=out, validate=False), reps=2)
    print(f"nrndm {wide.shape[1]:5d}  cells {C:6d}  {ms:8.2f} ms  {ms / C * 1e3:6.3f} us per cell")
