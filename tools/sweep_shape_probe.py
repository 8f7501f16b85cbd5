"""One block K-SVD sweep (three schedules: merged lazy = the default, lazy with two launches per block, eager) against the float64 C restatement over shapes that stress the kernel's capacity
limits: dense in-block coupling, single-block dictionaries, K not a multiple of the block, wide supports, wide signals.
usage: python tools/sweep_shape_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lyssandra_amd import engine as eng
from oracle import c_oracle

SHAPES = [(16, 16, 4, 200000), (8, 8, 3, 100000), (128, 128, 20, 100000), (200, 64, 8, 100000), (64, 1024, 32, 50000),
          (64, 100, 5, 300000), (64, 1030, 10, 100000), (256, 40, 6, 150000), (100, 24, 12, 120000), (64, 2048, 10, 1 << 19)]


def run(n, K, k, N, lazy):
    os.environ["LYS_BKSVD_LAZY"] = "0" if lazy == "0" else "1"
    os.environ["LYS_BKSVD_MERGED"] = "0" if lazy == "1u" else "1"   # "1u": lazy, X and Y as launches of their own
    gen = torch.Generator(device="cuda").manual_seed(11)
    Dt = torch.randn((n, K), device="cuda", generator=gen)
    Dt = Dt / Dt.norm(dim=0, keepdim=True)
    Xs = torch.randn((N, n), device="cuda", generator=gen)
    dd = eng.DeviceDictionary(n, K)
    dd.set(Dt)
    idx, coef, nnz = eng.bomp_encode(Xs, dd, k)
    h = (idx.cpu().numpy(), coef.double().cpu().numpy(), nnz.cpu().numpy())
    D0 = dd.D[:K, :n].t().contiguous().double().cpu().numpy()
    X = Xs.t().contiguous().double().cpu().numpy()
    R, _ = eng.residual(Xs, dd, idx, coef, nnz)
    unused = eng.ksvd_cycle(R, dd, idx, coef, nnz, buffers={})
    R2, _ = eng.residual(Xs, dd, idx, coef, nnz)
    drift = (R[:, :n] - R2[:, :n]).abs().max().item()
    Do, co, uo, _ = c_oracle.approx_ksvd_sparse(X, D0, *h, n_cycles=1)
    Dg = dd.to_host()
    ae = np.max(np.linalg.norm(Dg - Do, axis=0) / np.maximum(np.linalg.norm(Do, axis=0), 1e-30))
    ce = np.max(np.abs(coef.double().cpu().numpy() - co)) / np.abs(co).max()
    ok = ae < 1e-5 and ce < 1e-5 and drift < 1e-4 and unused == uo
    print("%s lazy=%s n=%d K=%d k=%d N=%d: atom err %.3g code err %.3g drift %.3g unused %d/%d"
          % ("ok  " if ok else "FAIL", lazy, n, K, k, N, ae, ce, drift, len(unused), len(uo)), flush=True)
    return ok


bad = 0
for sh in SHAPES:
    for lazy in ("1", "1u", "0"):
        try:
            bad += 0 if run(*sh, lazy) else 1
        except Exception as e:  # noqa: BLE001
            print("EXC ", sh, lazy, str(e)[:200], flush=True)
            bad += 1
print("failures:", bad)
