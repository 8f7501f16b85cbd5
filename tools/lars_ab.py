"""LARS coder at configs[3]'s shape (unit-norm 128-dim descriptors, 8192 atoms, mini-batch 32768, lambda 0.2): time per call
and KKT check, for the row-cache variants (LYS_LARS_CACHE = 0 | 4 | 6 | 8 -- read once per process: run once per value)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lyssandra_amd import engine

n, K, B, lam = 128, 8192, 32768, 0.2
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(4)
Xs = torch.randn((B, n), device=dev, generator=g)
Xs = Xs / Xs.norm(dim=1, keepdim=True)
D = torch.randn((n, K), device=dev, generator=g)
dd = engine.DeviceDictionary(n, K, dev)
dd.set(D / D.norm(dim=0, keepdim=True))
dd.gram()
ts = []
for it in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    idx, coef, nnz, steps, br = engine.lasso_encode(Xs, dd, lam, return_steps=True, solver='lars', return_breakpoints=True)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
# KKT in float64 on a sample: |d_j'(x - D a)| <= lambda (+tol), = lambda on the support
S = 512
Dh = dd.D[:K, :n].double()                       # [K][n]
a = torch.zeros((S, K), dtype=torch.float64, device=dev)
ii, cc, zz = idx[:S].long(), coef[:S].double(), nnz[:S]
for s in range(S):
    m = int(zz[s])
    a[s, ii[s, :m]] = cc[s, :m]
r = Xs[:S].double() - a @ Dh
corr = r @ Dh.t()
viol = (corr.abs() - lam).clamp_min(0).max().item()
on = (a != 0)
eq = ((corr.abs() - lam).abs() * on).max().item()
print("LYS_LARS_CACHE=%s: %.2f ms per mini-batch (runs %s) | mean nnz %.1f, breakpoints %.1f | KKT violation %.2e, |corr|-lambda on support %.2e"
      % (os.environ.get("LYS_LARS_CACHE", "default"), sorted(ts[1:])[1], ["%.1f" % t for t in ts], nnz.float().mean().item(),
         br.float().mean().item(), viol, eq))
