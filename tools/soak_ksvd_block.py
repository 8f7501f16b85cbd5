"""Soak of the block Gauss-Seidel sweep across its kernel templates: device sweep from the engine's own codes against the
float64 C restatement of `approx_ksvd` (oracle/bomp_oracle.c::lyso_approx_ksvd), several seeds per shape.

shapes: (n, K, k, N, cycles) -- FB = 1/2/4 feature blocks, B = 8 and B = 4 (n > 128), SL = 1/2 coefficient slots per lane
(k <= 16 / <= 32), ragged n, K not a multiple of the block size, two cycles."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lyssandra_amd import engine as eng
from oracle import c_oracle

SHAPES = [(64, 1024, 10, 1 << 19, 1), (64, 1000, 10, 1 << 18, 2), (128, 512, 8, 1 << 17, 1), (200, 256, 5, 1 << 16, 1),
          (64, 256, 20, 1 << 16, 1), (30, 128, 4, 1 << 15, 2), (100, 640, 12, 1 << 16, 1), (256, 512, 6, 1 << 15, 1)]
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 3


def atom_err(D, Dref):
    return np.max(np.linalg.norm(D - Dref, axis=0) / np.maximum(np.linalg.norm(Dref, axis=0), 1e-30))


worst = {"atom": 0.0, "code": 0.0, "err": 0.0}
for (n, K, k, N, cycles) in SHAPES:
    for seed in range(seeds):
        gen = torch.Generator(device="cuda").manual_seed(100 * n + seed)
        Dt = torch.randn((n, K), device="cuda", generator=gen)
        Dt = Dt / Dt.norm(dim=0, keepdim=True)
        Xs = torch.randn((N, n), device="cuda", generator=gen)
        dd = eng.DeviceDictionary(n, K)
        dd.set(Dt)
        idx, coef, nnz = eng.bomp_encode(Xs, dd, k)
        hi, hc, hn = idx.cpu().numpy(), coef.double().cpu().numpy(), nnz.cpu().numpy()
        D0 = dd.D[:K, :n].t().contiguous().double().cpu().numpy()
        X = Xs.t().contiguous().double().cpu().numpy()
        R, _ = eng.residual(Xs, dd, idx, coef, nnz)
        buf, unused = {}, []
        for _ in range(cycles):
            unused += eng.ksvd_cycle(R, dd, idx, coef, nnz, buffers=buf)
        err_dev = eng.approx_error(Xs, dd, idx, coef, nnz)
        Do, co, uo, err_o = c_oracle.approx_ksvd_sparse(X, D0, hi, hc, hn, n_cycles=cycles)
        ae = atom_err(dd.to_host(), Do)
        ce = np.max(np.abs(coef.double().cpu().numpy() - co)) / np.abs(co).max()
        ee = abs(err_dev - err_o) / err_o
        ok = unused == uo and ae < 1e-5 and ce < 1e-5 and ee < 1e-5
        worst = {"atom": max(worst["atom"], ae), "code": max(worst["code"], ce), "err": max(worst["err"], ee)}
        print("n=%3d K=%4d k=%2d N=%7d cycles=%d seed=%d: atom %.2e code %.2e error %.2e unused %d %s"
              % (n, K, k, N, cycles, seed, ae, ce, ee, len(unused), "ok" if ok else "FAIL"), flush=True)
        if not ok:
            sys.exit(1)
print("worst over %d runs: atom err %.2e, code err %.2e (of max|z|), error value %.2e relative"
      % (len(SHAPES) * seeds, worst["atom"], worst["code"], worst["err"]))
