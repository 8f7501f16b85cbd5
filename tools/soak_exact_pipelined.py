"""Soak of the pipelined exact K-SVD sweep (n <= 64, k <= 16: exact_k1_kernel / exact_k2_kernel, csrc/ksvd.hip) against the
float64 oracle's exact-SVD update in the reference's Gauss-Seidel order (ksvd.py:28-43): two larger shapes than the tests hold
and a run of random shapes (random n, K, k, N, random sets of unused atoms, coherent and incoherent dictionaries).
Usage: python tools/soak_exact_pipelined.py [n_random]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lyssandra_amd.dict_learning.ksvd import ksvd
from oracle import lyssa_oracle as orc


def one(rs, n, K, k, N, unused, noise):
    live = np.array([a for a in range(K) if a not in unused])
    Dt = rs.randn(n, K)
    Dt /= np.linalg.norm(Dt, axis=0)
    D0 = Dt + 0.3 * rs.randn(n, K)
    D0 = (D0 / np.linalg.norm(D0, axis=0)).astype(np.float32).astype(np.float64)
    kk = min(k, len(live))
    Z = np.zeros((K, N))
    for i in range(N):
        Z[rs.choice(live, kk, replace=False), i] = rs.randn(kk) + np.sign(rs.randn(kk))
    Z = Z.astype(np.float32).astype(np.float64)
    X = (Dt @ Z + noise * rs.randn(n, N)).astype(np.float32).astype(np.float64)
    Do, Zo, uo = orc.ksvd_exact(X, D0.copy(), Z.copy())
    Dh, Zh = D0.copy(), Z.copy()
    _, _, uh = ksvd(X, Dh, Zh, verbose=False)
    sgn = np.sign(np.sum(Dh * Do, axis=0))
    sgn[sgn == 0] = 1
    aerr = np.max(np.linalg.norm(Dh * sgn - Do, axis=0))
    zerr = np.abs(Zh * sgn[:, None] - Zo).max() / np.abs(Zo).max()
    same = list(uh) == list(uo) and np.array_equal(Zh != 0, Zo != 0)
    return aerr, zerr, same


rs = np.random.RandomState(int(os.environ.get("SOAK_SEED", "11")))
worst_a = worst_z = 0.0
ok = True
for (n, K, k, N) in [(64, 256, 6, 60000), (64, 1024, 10, 200000)]:
    a, z, same = one(rs, n, K, k, N, (), 0.1)
    print("n=%d K=%d k=%d N=%d: worst atom error %.3g, worst code error %.3g (rel. to max|z|), unused / pattern equal: %s"
          % (n, K, k, N, a, z, same))
    worst_a, worst_z, ok = max(worst_a, a), max(worst_z, z), ok and same
nr = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for t in range(nr):
    n = int(rs.randint(4, 65))
    K = int(rs.randint(1, int(os.environ.get("SOAK_KMAX", "200"))))
    k = int(rs.randint(1, min(16, K) + 1))
    N = int(rs.randint(50, 30000))
    unused = tuple(sorted(set(int(a) for a in rs.choice(K, rs.randint(0, max(1, K // 4)), replace=False)))) if K > k + 2 else ()
    if K - len(unused) < max(2, k):
        unused = ()
    a, z, same = one(rs, n, K, k, N, unused, float(rs.choice([0.02, 0.1, 0.5])))
    flag = "" if (a < 5e-5 and z < 5e-5 and same) else "   <-- CHECK"
    print("random %2d: n=%2d K=%3d k=%2d N=%5d unused=%2d: atom %.3g code %.3g equal %s%s" % (t, n, K, k, N, len(unused), a, z, same, flag))
    worst_a, worst_z, ok = max(worst_a, a), max(worst_z, z), ok and same
print("worst atom error %.3g, worst code error %.3g, unused lists and non-zero patterns equal everywhere: %s" % (worst_a, worst_z, ok))
