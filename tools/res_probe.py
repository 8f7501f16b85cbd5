"""What bounds the residual kernel (R = X - D Z, 2^20 64-dim patches, 1024 atoms): a device copy of the same bytes, then the kernel
at k = 1, 2, 5, 10 -- the slope is the price of one gathered dictionary row per signal.  usage: python tools/res_probe.py"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lyssandra_amd import engine as eng
n, K, N = 64, 1024, 1 << 20
g = torch.Generator(device="cuda").manual_seed(1)
Xs = torch.randn((N, n), device="cuda", generator=g)
dd = eng.DeviceDictionary(n, K)
dd.set((Xs[:K] / Xs[:K].norm(dim=1, keepdim=True)).t().contiguous())
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
Y = torch.empty_like(Xs)
print("torch copy 268 MB -> 268 MB: %.3f ms" % timeit(lambda: Y.copy_(Xs)))
for k in (1, 2, 5, 10):
    idx, coef, nnz = eng.bomp_encode(Xs, dd, k)
    R = None
    def f():
        global R
        R, _ = eng.residual(Xs, dd, idx, coef, nnz, want_R=True, want_err=False, out=R)
    print("residual k=%d: %.3f ms" % (k, timeit(f)))
