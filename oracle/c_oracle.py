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
        if not os.path.exists(LIB) or (os.path.exists(SRC) and os.path.getmtime(SRC) > os.path.getmtime(LIB)):
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


def approx_ksvd_sparse(X, D, idx, coef, nnz, n_cycles=1):
    """float64 approx K-SVD sweep on the sparse triplet (lyssa/dict_learning/ksvd.py:98-126).

    X (n, N), D (n, K) -> (D_new (n, K), coef_new [N, k], unused list, error).  Inputs are not modified."""
    lib = load()
    X = np.asarray(X, dtype=np.float64)
    n, N = X.shape
    K = np.asarray(D).shape[1]
    Xs = np.ascontiguousarray(X.T)
    Da = np.ascontiguousarray(np.asarray(D, dtype=np.float64).T)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    coef = np.array(coef, dtype=np.float64, order='C', copy=True)
    nnz = np.ascontiguousarray(nnz, dtype=np.int32)
    k = idx.shape[1]
    unused = np.empty(max(1, K * n_cycles), dtype=np.int32)
    err = ctypes.c_double(0.0)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    lib.lyso_approx_ksvd.restype = ctypes.c_int
    nu = lib.lyso_approx_ksvd(P(Xs), P(Da), ctypes.c_int(n), ctypes.c_int(K), ctypes.c_int(k), ctypes.c_int64(N),
                              P(idx), P(coef), P(nnz), ctypes.c_int(n_cycles), P(unused), ctypes.byref(err))
    return np.ascontiguousarray(Da.T), coef, unused[:nu].tolist(), float(err.value)


def synth_signals(seed, first, N, n):
    """Host counterpart of the engine's lys_synth_signals: fp32 array [N, n] (signal-major) of the synthetic patches
    first .. first + N - 1 (Philox4x32-10 + Box-Muller, same values as generated on the device)."""
    lib = load()
    X = np.empty((int(N), int(n)), dtype=np.float32)
    lib.lyso_synth_signals.restype = None
    lib.lyso_synth_signals(ctypes.c_uint64(int(seed)), ctypes.c_int64(int(first)), ctypes.c_int64(int(N)), ctypes.c_int(int(n)),
                           X.ctypes.data_as(ctypes.c_void_p))
    return X
