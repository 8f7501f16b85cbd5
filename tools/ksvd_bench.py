"""Config-2 timing (approx K-SVD, 1M 8x8 patches, 1024 atoms, k=10): per-stage times of one alternation on one GPU."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lyssandra_amd import engine

n, K, k = 64, 1024, 10
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
exact = len(sys.argv) > 3 and sys.argv[3] == "exact"   # exact rank-1 update (ksvd.py:19-43) instead of the approximate one
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
Xs = torch.randn((N, n), device=dev, generator=g)
dd = engine.DeviceDictionary(n, K, dev)
dd.set((Xs[:K] / Xs[:K].norm(dim=1, keepdim=True)).t().contiguous())   # D0 = first K signals, normalised


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    return r, (time.perf_counter() - t0) * 1e3


out, R, buffers = None, None, {}
for it in range(iters):
    out, t_enc = timed(lambda: engine.bomp_encode(Xs, dd, k, out=out))
    idx, coef, nnz = out
    (R, _), t_res = timed(lambda: engine.residual(Xs, dd, idx, coef, nnz, want_R=True, want_err=False, out=R))
    cycle = engine.ksvd_exact_cycle if exact else engine.ksvd_cycle
    unused, t_sweep = timed(lambda: cycle(R, dd, idx, coef, nnz, buffers=buffers))
    err, t_err = timed(lambda: engine.approx_error(Xs, dd, idx, coef, nnz))
    nnz_tot = int(nnz.sum().item())
    gbs = 8 * n * nnz_tot / (t_sweep * 1e-3) / 1e9          # 8n bytes per non-zero, the model bench.py prices with
    print("it %d: encode %.2f ms | residual %.2f ms | csr+sweep %.2f ms (%.0f GB/s by SURVEY 8(d)'s 8n bytes per non-zero, "
          "%.1f%% of 8 TB/s) | error %.2f ms | err=%.6g unused=%d"
          % (it, t_enc, t_res, t_sweep, gbs, gbs / 80.0, t_err, err, len(unused)))
