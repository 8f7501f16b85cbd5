"""Core clock in 0.4-ms slices through one encode (alpha0 GEMM ~1.1 ms, then the greedy kernel), once in a loop of back-to-back
encodes and once right behind a K-SVD sweep: ten clock probes (lys_debug_clock_probe) queued on a side stream beside the call."""
import ctypes
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lyssandra_amd import engine, _lib

lib = _lib.load()
n, K, k, N = 64, 1024, 10, 1 << 20
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
Xs = torch.randn((N, n), device=dev, generator=g)
dd = engine.DeviceDictionary(n, K, dev)
dd.set((Xs[:K] / Xs[:K].norm(dim=1, keepdim=True)).t().contiguous())
side = torch.cuda.Stream()
NP = 14
bufs = torch.zeros((NP, 2), dtype=torch.int64, device=dev)


def series(fn):
    torch.cuda.synchronize()
    for i in range(NP):
        _lib.check(lib.lys_debug_clock_probe(ctypes.c_void_p(bufs[i].data_ptr()), 400, ctypes.c_void_p(side.cuda_stream)), "probe")
    t0 = time.perf_counter()
    fn()
    torch.cuda.current_stream().synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    t = bufs.cpu().tolist()
    return ms, [100.0 * a / b for a, b in t]


out = engine.bomp_encode(Xs, dd, k)
idx, coef, nnz = out
R, buffers = None, {}
for _ in range(3):
    out = engine.bomp_encode(Xs, dd, k, out=out)
ms, cl = series(lambda: engine.bomp_encode(Xs, dd, k, out=out))
print("back to back : %.2f ms, MHz per 0.4 ms: %s" % (ms, " ".join("%.0f" % c for c in cl)))
for rep in range(3):
    R, _ = engine.residual(Xs, dd, idx, coef, nnz, want_R=True, want_err=False, out=R)
    engine.ksvd_cycle(R, dd, idx, coef, nnz, buffers=buffers)
    ms, cl = series(lambda: engine.bomp_encode(Xs, dd, k, out=out))
    print("behind sweep : %.2f ms, MHz per 0.4 ms: %s" % (ms, " ".join("%.0f" % c for c in cl)))
    ms, cl = series(lambda: engine.bomp_encode(Xs, dd, k, out=out))
    print("  and again  : %.2f ms, MHz per 0.4 ms: %s" % (ms, " ".join("%.0f" % c for c in cl)))
