"""LARS-lasso kernel vs the float64 oracle / KKT on a range of regimes (sparse .. dense supports), with timings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lyssandra_amd import engine
from oracle import lyssa_oracle as orc


def problem(seed, n, K, N, unit=True):
    rs = np.random.RandomState(seed)
    D = rs.randn(n, K)
    D /= np.linalg.norm(D, axis=0)
    if not unit:
        D *= rs.uniform(0.7, 1.4, size=K)[None, :]
    X = rs.randn(n, N)
    X /= np.linalg.norm(X, axis=0)
    return D.astype(np.float32).astype(np.float64), X.astype(np.float32).astype(np.float64)


for (n, K, N, lam, unit) in [(64, 256, 200, 0.15, True), (64, 1024, 200, 0.2, True), (64, 1024, 100, 0.02, True),
                             (32, 512, 100, 0.01, True), (128, 2048, 64, 0.15, True), (100, 6000, 16, 0.2, False),
                             (20, 40, 50, 0.1, True), (128, 8192, 32, 0.05, True)]:
    D, X = problem(n + K, n, K, N, unit)
    Xs = engine.signals_to_device(X)
    dd = engine.DeviceDictionary.from_host(D)
    res = {}
    for solver in ("cd", "lars"):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = engine.lasso_encode(Xs, dd, lam, return_steps=True, solver=solver, return_breakpoints=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        idx, coef, nnz, steps, br = out
        Z = orc.densify(idx.cpu().numpy(), coef.double().cpu().numpy(), nnz.cpu().numpy(), K)
        res[solver] = (Z, steps.cpu().numpy(), None if br is None else br.cpu().numpy(), dt)
    Zo = orc.lasso_encode(X, D, lam) if K <= 2048 else None
    line = "n=%d K=%d lam=%g nnz(mean/max)=%.1f/%d |" % (n, K, lam, (res["lars"][0] != 0).sum(0).mean(), (res["lars"][0] != 0).sum(0).max())
    for solver in ("cd", "lars"):
        Z, st, br, dt = res[solver]
        kkt = orc.lasso_kkt_violation(X, D, Z, lam)
        err = (np.max(np.abs(Z - Zo)) / np.abs(Zo).max()) if Zo is not None else float("nan")
        line += " %s: KKT %.1e err %.1e steps(max) %d%s %.2f ms |" % (
            solver, kkt, err, np.abs(st).max(), (" breaks(max) %d" % br.max()) if br is not None else "", dt * 1e3)
    print(line)
