"""One approximate K-SVD sweep at a larger size than the golden fixtures (N = 60 000, K = 256, k = 6, n = 64): device sweep
against the float64 oracle started from the SAME codes (the engine's), atoms / codes / error compared."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lyssandra_amd import engine as eng
from lyssandra_amd.dict_learning.ksvd import approx_ksvd
from oracle import lyssa_oracle as orc

rs = np.random.RandomState(3)
n, K, k, N = 64, 256, 6, 60000
Dt = rs.randn(n, K)
Dt /= np.linalg.norm(Dt, axis=0)
X = np.zeros((n, N))
for i in range(N):
    X[:, i] = Dt[:, rs.choice(K, k, replace=False)] @ rs.randn(k)
X = (X + 0.1 * rs.randn(n, N)).astype(np.float32).astype(np.float64)
D0 = (Dt + 0.4 * rs.randn(n, K))
D0 = (D0 / np.linalg.norm(D0, axis=0)).astype(np.float32).astype(np.float64)
Xs = eng.signals_to_device(X)
dd = eng.DeviceDictionary.from_host(D0)
Z0 = eng.densify(*eng.bomp_encode(Xs, dd, k), K)
for cycles in (1, 2):
    Do, Zo, uo = orc.approx_ksvd(X, D0.copy(), Z0.copy(), n_cycles=cycles)
    Dh, Zh = D0.copy(), Z0.copy()
    _, _, uh = approx_ksvd(X, Dh, Zh, n_cycles=cycles, verbose=False)
    aerr = np.max(np.linalg.norm(Dh - Do, axis=0))
    zerr = np.abs(Zh - Zo).max() / np.abs(Zo).max()
    e_o, e_h = np.sum((X - Do @ Zo) ** 2), np.sum((X - Dh @ Zh) ** 2)
    print("n_cycles=%d: worst atom error %.3g, worst code error %.3g (rel. to max|z|), error %.8g vs %.8g (rel. diff %.2g), "
          "unused equal: %s" % (cycles, aerr, zerr, e_h, e_o, abs(e_h - e_o) / e_o, list(uh) == list(uo)))
