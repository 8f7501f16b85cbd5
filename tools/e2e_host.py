"""End-to-end time of the drop-in call with HOST arrays (float64 (n, N) in, dense float64 (K, N) out), i.e. including the
host conversions and PCIe copies that `bench.py`'s device-resident `value` excludes."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lyssandra_amd.sparse_coding import sparse_encoder

for name, n, K, k, N in [("C1", 64, 256, 5, 10000), ("M", 64, 1024, 10, 100000)]:
    rs = np.random.RandomState(0)
    D = rs.randn(n, K)
    D /= np.linalg.norm(D, axis=0)
    X = rs.randn(n, N)
    se = sparse_encoder(algorithm='bomp', params={'n_nonzero_coefs': k}, verbose=False)
    se.encode(X[:, :256], D)                                   # warm-up (library load, allocations)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        Z = se.encode(X, D)
        ts.append(time.perf_counter() - t0)
    t = sorted(ts)[len(ts) // 2]
    ts2 = []
    for _ in range(5):
        t0 = time.perf_counter()
        idx, coef, nnz = se.encode_sparse(X, D)
        import torch
        torch.cuda.synchronize()
        ts2.append(time.perf_counter() - t0)
    t2 = sorted(ts2)[len(ts2) // 2]
    print("%s n=%d K=%d k=%d N=%d: encode() host->dense float64 %.1f ms = %.2f M patches/s (%.0f MB result) | "
          "encode_sparse() host->device triplet %.1f ms = %.2f M patches/s"
          % (name, n, K, k, N, t * 1e3, N / t / 1e6, Z.nbytes / 1e6, t2 * 1e3, N / t2 / 1e6))
