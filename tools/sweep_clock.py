"""Core clock while the block K-SVD sweep runs (config-2 shape): clock probes (lys_debug_clock_probe, one wave spinning on a side
stream) queued beside a loop of sweeps, and beside a loop of encodes for comparison.  The probe wave takes a wave slot of
one CU, so the sweep it measures runs a little slower than alone; the CLOCK it reports is the chip's."""
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
NP = 10
bufs = torch.zeros((NP, 2), dtype=torch.int64, device=dev)
out = engine.bomp_encode(Xs, dd, k)
idx, coef, nnz = out
R, _ = engine.residual(Xs, dd, idx, coef, nnz, want_err=False)
buffers = {}


def series(fn, slice_us):
    torch.cuda.synchronize()
    for i in range(NP):
        _lib.check(lib.lys_debug_clock_probe(ctypes.c_void_p(bufs[i].data_ptr()), slice_us, ctypes.c_void_p(side.cuda_stream)), "probe")
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    return ms, [round(100.0 * a / b) for a, b in bufs.cpu().tolist()]


for _ in range(2):
    engine.ksvd_cycle(R, dd, idx, coef, nnz, buffers=buffers)
ms, clk = series(lambda: [engine.ksvd_cycle(R, dd, idx, coef, nnz, buffers=buffers) for _ in range(4)], 1500)
print("4 sweeps %.2f ms; core clock per 1.5-ms slice (MHz): %s" % (ms, clk))
ms, clk = series(lambda: [engine.bomp_encode(Xs, dd, k, out=out) for _ in range(4)], 1500)
print("4 encodes %.2f ms; core clock per 1.5-ms slice (MHz): %s" % (ms, clk))
