"""ctypes wrapper of oracle/bomp_oracle.c (TEST INFRASTRUCTURE ONLY; see the header of the C file)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "bomp_oracle.c")
LIB = os.path.join(HERE, "libbomp_oracle.so")
_lib = None


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(SRC) > os.path.getmtime(LIB):
        subprocess.run(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-o", LIB, SRC, "-lm"], check=True)
    return LIB


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = ctypes.CDLL(LIB)
    return _lib


def bomp_encode_sparse(X, D, k):
    """Same contract as lyssa_oracle.bomp_encode_sparse: X (n, N), D (n, K) -> idx, coef, nnz, gap."""
    lib = load()
    X = np.asarray(X, dtype=np.float64)
    D = np.asarray(D, dtype=np.float64)
    n, N = X.shape
    K = D.shape[1]
    Xs = np.ascontiguousarray(X.T)
    Da = np.ascontiguousarray(D.T)
    G = np.empty((K, K))
    idx = np.empty((N, k), dtype=np.int32)
    coef = np.empty((N, k))
    nnz = np.empty(N, dtype=np.int32)
    gap = np.empty(N)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    lib.lyso_gram(P(Da), ctypes.c_int(n), ctypes.c_int(K), P(G))
    lib.lyso_bomp(P(Xs), P(Da), P(G), ctypes.c_int(n), ctypes.c_int(K), ctypes.c_int(k), ctypes.c_int64(N), P(idx),
                  P(coef), P(nnz), P(gap))
    return idx, coef, nnz, gap
