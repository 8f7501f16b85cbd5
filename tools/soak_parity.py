"""Soak test: many seeds of 2^18 Gaussian patches at the metric shape, GPU supports / order / coefficients against the
float64 C oracle (same protocol as tests/test_gpu_parity.py::test_bomp_direct_parity_262144_signals)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lyssandra_amd import engine as eng
from oracle import c_oracle

n, K, k, N = 64, 1024, 10, 1 << 18
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
if len(sys.argv) > 3:   # other templates of the greedy kernel: soak_parity.py <seeds> <K> <k>
    K, k = int(sys.argv[2]), int(sys.argv[3])
tot = ties = tie_diff = bad = early = early_ref = 0   # early: signals the engine stopped before k atoms (NOISE_REL floor, re-selection)
worst = 0.0
for seed in range(seeds):
    gen = torch.Generator(device="cuda").manual_seed(1000 + seed)
    Dt = torch.randn((n, K), device="cuda", generator=gen)
    Dt = Dt / Dt.norm(dim=0, keepdim=True)
    Xs = torch.randn((N, n), device="cuda", generator=gen) * (1.0 + seed)      # also vary the signal scale
    dd = eng.DeviceDictionary(n, K)
    dd.set(Dt)
    idx, coef, nnz = [t.cpu().numpy() for t in eng.bomp_encode(Xs, dd, k)]
    D = dd.D[:K, :n].t().contiguous().double().cpu().numpy()
    X = Xs.t().contiguous().double().cpu().numpy()
    oi, oc, on, gap = c_oracle.bomp_encode_sparse(X, D, k)
    ok = gap >= 1e-5
    tot += N
    ties += int((~ok).sum())
    tie_diff += int((idx[~ok] != oi[~ok]).any(axis=1).sum())
    bad += int((idx[ok] != oi[ok]).any(axis=1).sum()) + int((nnz[ok] != on[ok]).sum())
    early += int((nnz < k).sum())        # the noise-floor stop is a semantic the reference does not have:
    early_ref += int((on < k).sum())     # Gaussian signals have no exactly representable member, so both counts must be 0
    scale = np.abs(oc).max(axis=1, keepdims=True)
    worst = max(worst, float(np.max((np.abs(coef - oc) / scale)[ok])))
    print("seed %d: no-tie mismatches so far %d, worst coef err %.3g" % (seed, bad, worst), flush=True)
print("TOTAL %d signals: %d tie signals (%d selected differently), %d no-tie support/order mismatches, worst coefficient "
      "error %.3g of max|z|; stopped before k atoms (noise floor NOISE_REL / re-selection): engine %d, float64 oracle %d"
      % (tot, ties, tie_diff, bad, worst, early, early_ref))
