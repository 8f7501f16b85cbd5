"""Within-process interleaved A/B timing of greedy-kernel variants (run on the GPU box)."""
import ctypes
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lyssandra_amd import _lib, engine

lib = _lib.load()
n, K, k = 64, 1024, 10
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
Dt = torch.randn((n, K), device=dev, generator=g)
Dt = Dt / Dt.norm(dim=0, keepdim=True)
Xs = torch.randn((N, n), device=dev, generator=g)
dd = engine.DeviceDictionary(n, K, dev)
dd.set(Dt)
G = dd.gram()
a0 = torch.empty((N, 1024), dtype=torch.float32, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
_lib.check(lib.lys_alpha0(P(Xs), Xs.stride(0), P(dd.D), n, K, N, P(a0), st))
idx = torch.empty((N, k), dtype=torch.int32, device=dev)
coef = torch.empty((N, k), dtype=torch.float32, device=dev)
nnz = torch.empty((N,), dtype=torch.int32, device=dev)
configs = [("v0 regs-only 2w/SIMD", 0, 0, 10), ("v4 NLDS=3 3w/SIMD", 4, 0, 10), ("v15 NLDS=3 alpha0 rows L2-hot (64 rows)", 15, 0, 10), ("v5 NLDS=2 3w/SIMD (product)", 5, 0, 10),
           ("v6 NLDS=3 hot G (8 rows)", 6, 0, 10), ("v8 NLDS=3 G rows & 255 (1MB)", 8, 0, 10),
           ("v9 NLDS=3 G rows & 63 (256KB)", 9, 0, 10), ("v11 G rows & 511 (2MB)", 11, 0, 10),
           ("v10 G rows % 768 (3MB)", 10, 0, 10), ("v12 G rows % 896 (3.5MB)", 12, 0, 10), ("v1 hot G rows", 1, 0, 10), ("v2 no orth FMAs", 2, 0, 10),
           ("v3 ieee sqrt/div", 3, 0, 10), ("v0 lds 60K -> 2 blk/CU", 0, 60 * 1024, 10), ("v0 k=5", 0, 0, 5), ("v0 k=1", 0, 0, 1)]
times = {c[0]: [] for c in configs}
for rnd in range(8):
    for name, var, lds, kk in configs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.lys_debug_bomp_variant(P(a0), P(G), N, kk, P(idx), P(coef), P(nnz), var, lds, st))
        e1.record()
        torch.cuda.synchronize()
        if rnd >= 2:
            times[name].append(e0.elapsed_time(e1))
for name, *_ in configs:
    t = sorted(times[name])
    med = t[len(t) // 2]
    print("%-34s median %.4f ms  min %.4f ms  -> %.1f M sig/s" % (name, med, t[0], N / med / 1e3))
# v4 / v5 must reproduce v0 bit for bit
outs = []
for var in (0, 4, 5):
    _lib.check(lib.lys_debug_bomp_variant(P(a0), P(G), N, 10, P(idx), P(coef), P(nnz), var, 0, st))
    torch.cuda.synchronize()
    outs.append((idx.clone(), coef.clone(), nnz.clone()))
for o in outs[1:]:
    print("variant equals v0:", torch.equal(o[0], outs[0][0]), torch.equal(o[1], outs[0][1]), torch.equal(o[2], outs[0][2]))
# GEMM alone
tt = []
for rnd in range(8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(lib.lys_alpha0(P(Xs), Xs.stride(0), P(dd.D), n, K, N, P(a0), st))
    e1.record()
    torch.cuda.synchronize()
    tt.append(e0.elapsed_time(e1))
tt.sort()
print("alpha0 GEMM median %.4f ms -> %.1f M sig/s, %.1f TFLOP/s" % (tt[4], N / tt[4] / 1e3, 2 * n * K * N / tt[4] / 1e9))
