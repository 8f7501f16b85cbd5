"""Dictionary-learning helpers (lyssa/dict_learning/utils.py), host control flow + device reductions."""
import numpy as np

from .. import engine
from ..utils.math import norm_cols


def init_dictionary(X, n_atoms, method='data', return_unused_data=False, normalize=True):
    """lyssa/dict_learning/utils.py:35-75, method='data' (and 'random').

    Host logic, global numpy RNG exactly like the reference (:60): candidates are the columns with energy
    > 1e-6, ``np.random.choice(len(idxs), n_atoms, replace=False)``, D = X[:, chosen] (a copy), optional
    norm_cols, ``unused_data`` = remaining candidate indices as a Python list.
    """
    X = np.asarray(X)
    if method == "data":
        energy = np.einsum('ij,ij->j', X, X)
        idxs = np.flatnonzero(energy > 1e-6).tolist()
        if len(idxs) < n_atoms:
            raise ValueError("not enough datapoints to initialize the dictionary")
        subset = np.random.choice(len(idxs), size=n_atoms, replace=False)
        subset_idxs = np.array(idxs).astype(int)[subset]
        D = np.array(X[:, subset_idxs], dtype=np.float64)
        if normalize:
            D = norm_cols(D)
        if return_unused_data:
            s = set(subset_idxs.tolist())
            return D, [x for x in idxs if x not in s]
        return D
    if method == "random":
        D = np.random.randn(X.shape[0], n_atoms)
        return norm_cols(D)
    raise NotImplementedError("init_dictionary(method=%r) is outside the accelerated path" % (method,))


def approx_error(D, Z, X, n_jobs=1):
    """lyssa/dict_learning/utils.py:14-19 -- ||X - DZ||_F^2, evaluated on the device from the sparse form of Z."""
    Xs = engine.signals_to_device(X)
    dd = engine.DeviceDictionary.from_host(D)
    idx, coef, nnz = engine.sparsify_host(Z)
    return engine.approx_error(Xs, dd, idx, coef, nnz)


def average_mutual_coherence(D):
    """lyssa/dict_learning/utils.py:7-11 -- mean off-diagonal |D'D| (Gram from the MFMA GEMM, reduction on the device)."""
    import ctypes
    from .. import _lib
    torch = engine.require_gpu()
    lib = _lib.load()
    D = np.asarray(D)
    dd = engine.DeviceDictionary.from_host(D)
    K = dd.K
    G = dd.gram()
    out = torch.zeros((1,), dtype=torch.float64, device=dd.device)
    _lib.check(lib.lys_offdiag_abs_sum(ctypes.c_void_p(G.data_ptr()), K, ctypes.c_void_p(out.data_ptr()),
                                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "lys_offdiag_abs_sum")
    return float(out.item()) / float(K * (K - 1))
