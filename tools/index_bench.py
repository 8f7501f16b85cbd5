"""Time of the block-sweep index build alone (lys_bksvd_index: count pass, scans, fill pass) at the config-2 shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lyssandra_amd import engine, _lib
n, K, k, N = 64, 1024, 10, 1 << 20
g = torch.Generator(device="cuda").manual_seed(3)
Xs = torch.randn((N, n), device="cuda", generator=g)
dd = engine.DeviceDictionary(n, K)
dd.set((Xs[:K] / Xs[:K].norm(dim=1, keepdim=True)).t().contiguous())
idx, coef, nnz = engine.bomp_encode(Xs, dd, k)
R, _ = engine.residual(Xs, dd, idx, coef, nnz, want_err=False)
ops = engine.HipBlockKsvdOps(R, dd, idx, coef, nnz, {})
ts = []
for it in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ops.begin()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("index build (begin(): index + slab memset): min %.3f ms, runs %s" % (min(ts[2:]), ["%.3f" % t for t in ts]))
