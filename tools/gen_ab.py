"""Greedy kernel time per shape (Kp, k): LYS_BOMP_GEN1=1 python tools/gen_ab.py vs python tools/gen_ab.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lyssandra_amd import _lib, engine
lib = _lib.load()
dev = torch.device("cuda", 0)
P = lambda t: ctypes.c_void_p(t.data_ptr())
SH = [(64, 256, 5, 1 << 20), (64, 256, 10, 1 << 20), (64, 512, 5, 1 << 20), (64, 512, 10, 1 << 20), (64, 1024, 5, 1 << 20), (64, 1024, 10, 1 << 20)]
if len(sys.argv) > 1:
    SH = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
for n, K, k, N in SH:
    g = torch.Generator(device=dev).manual_seed(1)
    Dt = torch.randn((n, K), device=dev, generator=g); Dt = Dt / Dt.norm(dim=0, keepdim=True)
    Xs = torch.randn((N, n), device=dev, generator=g)
    dd = engine.DeviceDictionary(n, K, dev); dd.set(Dt)
    G = dd.gram()
    a0 = torch.empty((N, K), dtype=torch.float32, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.lys_alpha0(P(Xs), Xs.stride(0), P(dd.D), n, K, N, P(a0), st))
    o = (torch.empty((N, k), dtype=torch.int32, device=dev), torch.empty((N, k), dtype=torch.float32, device=dev), torch.empty((N,), dtype=torch.int32, device=dev))
    ts = []
    for r in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.lys_bomp_from_alpha0(P(a0), P(G), K, k, N, P(o[0]), P(o[1]), P(o[2]), st))
        e1.record(); torch.cuda.synchronize()
        if r >= 2: ts.append(e0.elapsed_time(e1))
    ts.sort()
    print("K=%d k=%d: greedy median %.4f ms (%.0f M sig/s) checksum %d" % (K, k, ts[len(ts)//2], N / ts[len(ts)//2] / 1e3, int(o[0].sum().item())), flush=True)
