"""Why does the encode of a K-SVD alternation take 4.7 ms when the same call takes 4.0 ms in bench.py's step loop?
Times three encodes in a row behind each sweep (same dictionary, same signals), with a dictionary learned from the data and with a
random one, and the core clock beside each (lys_debug_clock_probe)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from lyssandra_amd import engine

n, K, k, N = 64, 1024, 10, 1 << 20
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
Xs = torch.randn((N, n), device=dev, generator=g)


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    return r, (time.perf_counter() - t0) * 1e3


for label in ("data", "random"):
    dd = engine.DeviceDictionary(n, K, dev)
    if label == "data":
        D0 = (Xs[:K] / Xs[:K].norm(dim=1, keepdim=True))
    else:
        D0 = torch.randn((K, n), device=dev, generator=g)
        D0 = D0 / D0.norm(dim=1, keepdim=True)
    dd.set(D0.t().contiguous())
    out, R, buffers = None, None, {}
    for it in range(4):
        ts, cl = [], []
        for rep in range(3):
            box = {}

            def enc():
                box["o"] = engine.bomp_encode(Xs, dd, k, out=out)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            c = bench.probe_sclk(enc, 3000)
            ts.append((time.perf_counter() - t0) * 1e3)
            cl.append(c)
            out = box["o"]
        idx, coef, nnz = out
        (R, _), t_res = timed(lambda: engine.residual(Xs, dd, idx, coef, nnz, want_R=True, want_err=False, out=R))
        _, t_sweep = timed(lambda: engine.ksvd_cycle(R, dd, idx, coef, nnz, buffers=buffers))
        print("%-6s it %d: encode x3 %.2f %.2f %.2f ms (core clock %.0f %.0f %.0f MHz) | sweep %.2f ms | mean nnz %.2f"
              % (label, it, ts[0], ts[1], ts[2], cl[0], cl[1], cl[2], t_sweep, float(nnz.float().mean().item())))

# the same encode, 20 calls enqueued back to back (one synchronisation at the end) against 20 calls with a synchronisation each
for label, sync_each in (("back-to-back", False), ("sync each", True)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        out = engine.bomp_encode(Xs, dd, k, out=out)
        if sync_each:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print("%s: %.3f ms per encode" % (label, (time.perf_counter() - t0) * 1e3 / 20))
# ... and with one sweep between the encodes, everything enqueued without a synchronisation
idx, coef, nnz = out
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    out = engine.bomp_encode(Xs, dd, k, out=out)
    R, _ = engine.residual(Xs, dd, idx, coef, nnz, want_R=True, want_err=False, out=R)
    engine.ksvd_cycle(R, dd, idx, coef, nnz, buffers=buffers)
torch.cuda.synchronize()
print("encode + residual + sweep, no synchronisation inside: %.3f ms per alternation" % ((time.perf_counter() - t0) * 1e3 / 10))
