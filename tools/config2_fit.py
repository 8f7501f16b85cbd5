"""configs[1] end to end through the drop-in class: ksvd_coder(approx=True).fit on 2^20 synthetic 8x8 patches (host float64
array in), 1024 atoms, k = 10, max_iter = 50 (the reference's patience quirk stops it after 11 iterations)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # imported (and the library loaded) before the clock starts: about 1 s of one-off start-up otherwise
from lyssandra_amd import _lib
from lyssandra_amd.dict_learning import ksvd_coder
from lyssandra_amd.sparse_coding import sparse_encoder

n, K, k, N = 64, 1024, 10, 1 << 20
rs = np.random.RandomState(0)
X = rs.randn(n, N)
se = sparse_encoder(algorithm='bomp', params={'n_nonzero_coefs': k}, verbose=False)
_lib.load()
torch.zeros(1, device="cuda")
np.random.seed(1)
kc = ksvd_coder(n_atoms=K, sparse_coder=se, max_iter=50, approx=True, verbose=False)
import cProfile
import pstats
pr = cProfile.Profile() if "--profile" in sys.argv else None
t0 = time.perf_counter()
if pr:
    pr.enable()
kc.fit(X)
torch.cuda.synchronize()
if pr:
    pr.disable()
t = time.perf_counter() - t0
if pr:
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
print("ksvd_coder.fit: N=%d K=%d k=%d max_iter=50 -> %.2f s wall (host float64 input, 11 alternations), D %s"
      % (N, K, k, t, kc.D.shape))
