"""Online-DL stage timing at a config-4-like shape (128-dim descriptors, 8192 atoms; Batch-OMP as the inner solver,
the reference's LARS is out of scope) and at the metric shape."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lyssandra_amd import engine

dev = torch.device("cuda", 0)


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    return r, (time.perf_counter() - t0) * 1e3


for name, n, K, k, B in [("metric n=64 K=1024 k=10 batch=65536", 64, 1024, 10, 65536),
                         ("config-4-like n=128 K=8192 k=10 batch=32768", 128, 8192, 10, 32768)]:
    g = torch.Generator(device=dev).manual_seed(2)
    Xs = torch.randn((3 * B, n), device=dev, generator=g)
    dd = engine.DeviceDictionary(n, K, dev)
    dd.set((Xs[:K] / Xs[:K].norm(dim=1, keepdim=True)).t().contiguous())
    state = engine.OdlState(dd)
    for b in range(3):
        Xb = Xs[b * B:(b + 1) * B]
        (idx, coef, nnz), t_enc = timed(lambda: engine.bomp_encode(Xb, dd, k))
        _, t_upd = timed(lambda: state.batch_update(Xb, idx, coef, nnz, 0.5 if b else 0.0))
        nrm = dd.D[:K, :n].norm(dim=1)
        print("%s | batch %d: encode %.2f ms (%.1f M patches/s) | stats+update %.2f ms | atom norms [%.6f, %.6f] A diag max %.3g"
              % (name, b, t_enc, B / t_enc / 1e3, t_upd, nrm.min().item(), nrm.max().item(),
                 state.A.diagonal().max().item()))
    del state, dd, Xs
    engine.release_workspaces()
    torch.cuda.empty_cache()

# configs[3] with the inner solver it names: LARS-lasso on unit-norm 128-dim descriptors, 8192 atoms, mini-batch 32768
n, K, B, lam = 128, 8192, 32768, 0.2
g = torch.Generator(device=dev).manual_seed(4)
Xs = torch.randn((3 * B, n), device=dev, generator=g)
Xs = Xs / Xs.norm(dim=1, keepdim=True)
dd = engine.DeviceDictionary(n, K, dev)
D0 = torch.randn((n, K), device=dev, generator=g)
dd.set(D0 / D0.norm(dim=0, keepdim=True))
state = engine.OdlState(dd)
for b in range(3):
    Xb = Xs[b * B:(b + 1) * B]
    (idx, coef, nnz, steps, br), t_enc = timed(lambda: engine.lasso_encode(Xb, dd, lam, return_steps=True, solver='lars',
                                                                          return_breakpoints=True))
    _, t_upd = timed(lambda: state.batch_update(Xb, idx, coef, nnz, 0.9 if b else 0.0))
    print("config 4 (LARS, lambda=%.2f) | batch %d: lasso %.2f ms (%.2f M signals/s, mean nnz %.1f, mean breakpoints %.1f) | "
          "stats+update %.2f ms" % (lam, b, t_enc, B / t_enc / 1e3, nnz.float().mean().item(), br.float().mean().item(), t_upd))
