"""Per-phase cycle breakdown of one wave's greedy chain (s_memtime stamps, lys_debug_bomp_variant 150+)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lyssandra_amd import _lib, engine
lib = _lib.load()
n, K, k = 64, 1024, 10
N = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
variants = [int(v) for v in sys.argv[2:]] or [150]
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
names = ["alpha0 wait", "argmax+reduce", "owner lookup", "extract+pivot", "update+commit", "backsubst"]
for v in variants:
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.lys_debug_bomp_variant(P(a0), P(G), N, k, P(idx), P(coef), P(nnz), v, 0, st))
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    c = coef[N // 8: N - N // 8].double()  # steady state: skip launch ramp and tail
    m = c.mean(dim=0)
    tot = float(m[:6].sum())
    print("variant %d: %.4f ms; s_memtime ticks per signal (mean over %d signals), total %.0f" % (v, ms, c.shape[0], tot))
    for i, nm in enumerate(names):
        print("   %-14s %9.0f  (%4.1f %%)%s" % (nm, float(m[i]), 100 * float(m[i]) / tot, "   per step %.0f" % (float(m[i]) / 10) if 1 <= i <= 4 else ""))
