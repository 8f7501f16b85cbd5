"""Soak of the working-set lasso coder (lasso_ws_kernel + its hand-over to the homotopy / plain coordinate descent) on random
shapes: KKT conditions in float64 (computed on the device with torch), objective against the pure homotopy (LYS_LASSO_WS=0), the
share of signals the pass solves, truncation / hand-over behaviour.  usage: soak_lasso_ws.py [n_shapes] [seed]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lyssandra_amd import engine

n_shapes = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda", 0)


def dense(idx, coef, nnz, K):
    N, kc = idx.shape
    a = torch.zeros((N, K), dtype=torch.float64, device=dev)
    valid = torch.arange(kc, device=dev)[None, :] < nnz[:, None]
    rows = torch.arange(N, device=dev)[:, None].expand(N, kc)
    a[rows[valid], idx[valid].long()] = coef[valid].double()
    return a


worst = 0.0
for it in range(n_shapes):
    n = int(rs.choice([16, 32, 64, 100, 128, 200]))
    K = int(rs.choice([1024, 1500, 2048, 3000, 4096, 8192, 16384]))
    N = int(rs.choice([64, 200, 777]))
    lam = float(rs.choice([0.3, 0.2, 0.15, 0.1, 0.05]))
    unit = bool(rs.rand() < 0.7)
    g = torch.Generator(device=dev).manual_seed(1000 + it)
    D = torch.randn((n, K), device=dev, generator=g)
    D = D / D.norm(dim=0, keepdim=True)
    if not unit:
        D = D * (0.7 + 0.7 * torch.rand((1, K), device=dev, generator=g))
    Xs = torch.randn((N, n), device=dev, generator=g)
    Xs = Xs / Xs.norm(dim=1, keepdim=True)
    dd = engine.DeviceDictionary(n, K, dev)
    dd.set(D)
    res = {}
    for ws in ("1", "0"):
        os.environ["LYS_LASSO_WS"] = ws
        idx, coef, nnz, steps, br = engine.lasso_encode(Xs, dd, lam, return_steps=True, solver='lars', return_breakpoints=True)
        a = dense(idx, coef, nnz, K)
        Dh = dd.D[:K, :n].double()
        r = Xs.double() - a @ Dh
        corr = r @ Dh.t()
        viol = (corr.abs() - lam).clamp_min(0).max().item()
        eq = ((corr.abs() - lam).abs() * (a != 0)).max().item()
        obj = 0.5 * r.pow(2).sum(1) + lam * a.abs().sum(1)
        res[ws] = (viol, eq, obj, nnz, steps, br)
    os.environ["LYS_LASSO_WS"] = "1"
    v1, e1, o1, nz1, st1, br1 = res["1"]
    v0, e0, o0, nz0, st0, br0 = res["0"]
    solved = int((br1 <= 0).sum().item())
    dobj = ((o1 - o0) / o0).max().item()
    worst = max(worst, v1, e1)
    print("n=%3d K=%5d N=%3d lam=%.2f %s: pass solved %3d / %3d | nnz mean %5.1f max %3d | KKT %.1e (homotopy %.1e) on-support %.1e | "
          "objective excess over the homotopy %.1e | steps min %d" % (n, K, N, lam, "unit" if unit else "free", solved, N, nz1.float().mean().item(),
                                                                     int(nz1.max().item()), v1, v0, e1, dobj, int(st1.min().item())))
    assert v1 < 2e-5 and e1 < 2e-5 and dobj < 1e-6 and int(st1.min().item()) >= 0, "soak failure"
print("worst KKT residual %.2e over %d shapes" % (worst, n_shapes))
