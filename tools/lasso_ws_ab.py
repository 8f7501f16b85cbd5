"""Working-set lasso (lasso_ws_kernel, the default of solver='cd' for K >= 1024) against the plain coordinate-descent kernel
(LYS_LASSO_WS=0) and the LARS homotopy: time per configs[3] mini-batch, KKT in float64 on a sample, agreement of the codes;
then accuracy regimes against the float64 oracle.  usage: lasso_ws_ab.py [quick]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lyssandra_amd import engine


def kkt64(Xs, dd, idx, coef, nnz, lam, S=512):
    n, K = dd.n, dd.K
    dev = Xs.device
    Dh = dd.D[:K, :n].double()
    a = torch.zeros((S, K), dtype=torch.float64, device=dev)
    ii, cc, zz = idx[:S].long(), coef[:S].double(), nnz[:S]
    for s in range(S):
        m = int(zz[s])
        a[s, ii[s, :m]] = cc[s, :m]
    corr = (Xs[:S].double() - a @ Dh) @ Dh.t()
    viol = (corr.abs() - lam).clamp_min(0).max().item()
    eq = ((corr.abs() - lam).abs() * (a != 0)).max().item()
    obj = (0.5 * (Xs[:S].double() - a @ Dh).pow(2).sum(1) + lam * a.abs().sum(1))
    return viol, eq, obj, a


def big(n=128, K=8192, B=32768, lam=0.2):
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(4)
    Xs = torch.randn((B, n), device=dev, generator=g)
    Xs = Xs / Xs.norm(dim=1, keepdim=True)
    D = torch.randn((n, K), device=dev, generator=g)
    dd = engine.DeviceDictionary(n, K, dev)
    dd.set(D / D.norm(dim=0, keepdim=True))
    dd.gram()
    res = {}
    for name, solver, env in (("working-set cd", "cd", "1"), ("lars + polish", "lars", "0"), ("plain cd", "cd", "0")):
        if name == "plain cd" and B > 4096:
            Xrun = Xs[:4096]
        else:
            Xrun = Xs
        os.environ["LYS_LASSO_WS"] = env
        ts = []
        for it in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = engine.lasso_encode(Xrun, dd, lam, return_steps=True, solver=solver)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        idx, coef, nnz, steps = out[:4]
        viol, eq, obj, a = kkt64(Xrun, dd, idx, coef, nnz, lam)
        res[name] = (a, obj)
        st = steps.cpu().numpy()
        print("%-15s n=%d K=%d B=%d lam=%g: %.2f ms (runs %s) | nnz mean %.1f max %d | steps mean %.0f max %d min %d | KKT viol %.2e, "
              "||corr|-lam| on support %.2e" % (name, n, K, Xrun.shape[0], lam, min(ts[1:]), ["%.1f" % x for x in ts], nnz.float().mean().item(),
                                               int(nnz.max().item()), st.mean(), st.max(), st.min(), viol, eq))
    os.environ["LYS_LASSO_WS"] = "1"
    a0, o0 = res["working-set cd"]
    for other in ("lars + polish", "plain cd"):
        a1, o1 = res[other]
        print("   vs %-14s: max |code diff| %.2e (rel. to max |code| %.3f) | objective diff max %.2e" %
              (other, (a0 - a1).abs().max().item(), a1.abs().max().item(), (o0 - o1).abs().max().item()))


def regimes():
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
    for (n, K, N, lam, unit) in [(64, 1024, 200, 0.2, True), (64, 1024, 100, 0.02, True), (128, 2048, 64, 0.15, True),
                                 (100, 6000, 16, 0.2, False), (128, 8192, 32, 0.05, True), (16, 1024, 64, 0.001, True),
                                 (200, 2048, 32, 0.01, True)]:
        D, X = problem(n + K, n, K, N, unit)
        Xs = engine.signals_to_device(X)
        dd = engine.DeviceDictionary.from_host(D)
        line = "n=%d K=%d lam=%g |" % (n, K, lam)
        Zs = {}
        for env in ("1", "0"):
            os.environ["LYS_LASSO_WS"] = env
            idx, coef, nnz, steps = engine.lasso_encode(Xs, dd, lam, return_steps=True)
            Z = orc.densify(idx.cpu().numpy(), coef.double().cpu().numpy(), nnz.cpu().numpy(), K)
            Zs[env] = Z
            st = steps.cpu().numpy()
            line += " %s: nnz mean %.1f max %d, KKT %.1e, steps max %d min %d |" % (
                "ws" if env == "1" else "plain", (Z != 0).sum(0).mean(), (Z != 0).sum(0).max(), orc.lasso_kkt_violation(X, D, Z, lam), st.max(), st.min())
        os.environ["LYS_LASSO_WS"] = "1"
        line += " ws vs plain: %.1e" % (np.abs(Zs["1"] - Zs["0"]).max() / max(np.abs(Zs["0"]).max(), 1e-30))
        if K <= 2048:
            Zo = orc.lasso_encode(X, D, lam)
            line += " | ws vs float64 oracle: %.1e" % (np.abs(Zs["1"] - Zo).max() / np.abs(Zo).max())
        print(line)


if __name__ == "__main__":
    big()
    if len(sys.argv) < 2:
        regimes()
