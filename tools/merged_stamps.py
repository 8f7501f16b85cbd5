import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lyssandra_amd import engine, _lib
n, K, k, N = 64, 1024, 10, 1 << 20
g = torch.Generator(device="cuda").manual_seed(3)
Xs = torch.randn((N, n), device="cuda", generator=g)
dd = engine.DeviceDictionary(n, K)
dd.set((Xs[:K] / Xs[:K].norm(dim=1, keepdim=True)).t().contiguous())
idx, coef, nnz = engine.bomp_encode(Xs, dd, k)
R, _ = engine.residual(Xs, dd, idx, coef, nnz, want_err=False)
ops = engine.HipBlockKsvdOps(R, dd, idx, coef, nnz, {})
ops.begin()
lib = _lib.load()
out = np.zeros(64, dtype=np.uint64)
ops.step(0, 0)
for c in range(1, 92):
    ops.step(3, c)
    if c in (40, 41, 90, 91):
        torch.cuda.synchronize()
        lib.lys_debug_timestamps(out.ctypes.data_as(ctypes.c_void_p))
        w = out.astype(np.int64)
        b = w[0]
        f = lambda i: (w[i] - b) / 100.0 if w[i] else float('nan')
        print("merged(%d): narrow 0.00 .. atoms done %.2f .. end %.2f | Y part of wg0: start %.2f, LDS+sync %.2f, collect/drain done %.2f, flushed %.2f, end %.2f | wg/2: start %.2f, LDS+sync %.2f, drain done %.2f, flushed %.2f, end %.2f"
              % (c, f(4), f(5), f(32), f(33), f(35), f(36), f(37), f(48), f(49), f(51), f(52), f(53)))
