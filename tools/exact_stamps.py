"""Phase timestamps of the exact K-SVD kernels (last atom of a sweep at config-2 size)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import numpy as np
import torch
from lyssandra_amd import engine, _lib

n, K, k, N = 64, 1024, 10, 1 << 20
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
Xs = torch.randn((N, n), device=dev, generator=g)
dd = engine.DeviceDictionary(n, K, dev)
dd.set((Xs[:K] / Xs[:K].norm(dim=1, keepdim=True)).t().contiguous())
idx, coef, nnz = engine.bomp_encode(Xs, dd, k)
R, _ = engine.residual(Xs, dd, idx, coef, nnz, want_R=True, want_err=False)
buf = {}
for _ in range(2):
    engine.ksvd_exact_cycle(R, dd, idx, coef, nnz, buffers=buf)
torch.cuda.synchronize()
out = (ctypes.c_uint64 * 64)()
_lib.check(_lib.load().lys_debug_timestamps(out), "stamps")
t = np.array(list(out), dtype=np.float64)
e = t[16:24]
print("eig : C loaded %.2f | lanczos %.2f | end %.2f us | steps %d" % ((e[1] - e[0]) / 100, (e[2] - e[0]) / 100, (e[3] - e[0]) / 100, int(e[4])))
q = t[24:32]
if e[5] > e[0]:
    print("eig steps (build with -DLYS_EXACT_STEP_STAMPS): " + " | ".join("%.2f" % ((x - e[0]) / 100) for x in e[5:8]))
if e[14 - 8 + 8 - 8] >= 0 and t[16 + 14] > 0:
    print("eig : mean Lanczos steps %.3f over %d solves" % (t[16 + 13] / t[16 + 14], int(t[16 + 14])))
if os.environ.get("K1_STAMPS"):
    print("K1 roles (build with -DLYS_EXACT_K1_STAMPS), in-kernel us of the first workgroup of each: reduce %.2f | shared rows %.2f | apply %.2f"
          % (e[5] / 100, e[6] / 100, e[7] / 100))
print("gram: descriptors %.2f | rounds %.2f | end %.2f us | signals in wg0 %d" % ((q[1] - q[0]) / 100, (q[2] - q[0]) / 100, (q[3] - q[0]) / 100, int(q[4])))
