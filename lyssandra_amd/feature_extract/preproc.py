"""Per-patch preprocessing on the device -- lyssa/feature_extract/preproc.py:46-80.

Per datapoint (one wave per signal, fused scale / centre / normalise): 'scaling', 'local_centering',
'contrast_normalization', 'normalization'.  Per FEATURE over the whole dataset: 'global_centering',
'global_standarization' (fp64 column statistics + one affine pass) and ZCA 'whitening' (`zca_transform`, :18-31:
X'X with fp32 MFMA partial sums accumulated in fp64, the n x n eigen-decomposition on the host with scipy `eigh` like
the reference, the transform X W as one MFMA GEMM).
"""
import ctypes

import numpy as np

from .. import _lib, engine
from ..utils.math import normalize as _normalize

_FLAGS = {  # name -> (scale, center, normalize)
    'scaling': (1.0 / 255.0, False, False),
    'local_centering': (1.0, True, False),
    'contrast_normalization': (1.0, True, True),
    'normalization': (1.0, False, True),
}


def _feature_mean_std(Xs):
    """Per-feature mean and (population) std over all signals, float64 host vectors."""
    torch = engine.require_gpu()
    lib = _lib.load()
    N, n = int(Xs.shape[0]), int(Xs.shape[1])
    st = torch.zeros((2, n), dtype=torch.float64, device=Xs.device)
    _lib.check(lib.lys_feature_stats(ctypes.c_void_p(Xs.data_ptr()), engine._ld(Xs), n, N,
                                     ctypes.c_void_p(st[0].data_ptr()), ctypes.c_void_p(st[1].data_ptr()),
                                     ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "lys_feature_stats")
    h = st.cpu().numpy()
    mean = h[0] / max(N, 1)
    var = np.maximum(h[1] / max(N, 1) - mean * mean, 0.0)
    return mean, np.sqrt(var)


def _feature_affine(Xs, shift, scale):
    torch = engine.require_gpu()
    lib = _lib.load()
    sh = torch.from_numpy(np.asarray(shift, dtype=np.float32)).to(Xs.device)
    sc = torch.from_numpy(np.asarray(scale, dtype=np.float32)).to(Xs.device)
    _lib.check(lib.lys_feature_affine(ctypes.c_void_p(Xs.data_ptr()), engine._ld(Xs), int(Xs.shape[1]), int(Xs.shape[0]),
                                      ctypes.c_void_p(sh.data_ptr()), ctypes.c_void_p(sc.data_ptr()),
                                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "lys_feature_affine")
    return Xs


def zca_whiten_device(Xs, bias=0.1):
    """`zca_transform` (feature_extract/preproc.py:18-31) on a signal-major fp32 cuda tensor [N, n]: centre every
    feature, C = X'X / N + bias I, W = V diag(eigs^-1/2) V', return X W (a new tensor)."""
    from scipy.linalg import eigh
    torch = engine.require_gpu()
    lib = _lib.load()
    N, n = int(Xs.shape[0]), int(Xs.shape[1])
    mean, _ = _feature_mean_std(Xs)
    _feature_affine(Xs, mean, np.ones(n))
    C = torch.zeros((n, n), dtype=torch.float64, device=Xs.device)
    _lib.check(lib.lys_covariance(ctypes.c_void_p(Xs.data_ptr()), engine._ld(Xs), n, N, ctypes.c_void_p(C.data_ptr()),
                                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "lys_covariance")
    eigs, eigv = eigh(C.cpu().numpy() / N + bias * np.identity(n))
    W = np.dot(eigv * np.sqrt(1.0 / eigs), eigv.T)            # symmetric
    # X W through the engine's GEMM: W plays the dictionary (its columns are the "atoms"), alpha0 = X W
    dd = engine.DeviceDictionary.from_host(W, Xs.device)
    out = torch.empty((N, dd.Kp), dtype=torch.float32, device=Xs.device)
    if N > 0:
        _lib.check(lib.lys_alpha0(ctypes.c_void_p(Xs.data_ptr()), engine._ld(Xs), ctypes.c_void_p(dd.D.data_ptr()), n, n, N,
                                  ctypes.c_void_p(out.data_ptr()),
                                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "lys_alpha0")
    return out[:, :n]


def preproc_device(Xs, name):
    """On a signal-major fp32 cuda tensor [N, n]; in place except 'whitening', which returns a new tensor."""
    torch = engine.require_gpu()
    lib = _lib.load()
    if name in _FLAGS:
        scale, center, norm = _FLAGS[name]
        _lib.check(lib.lys_preproc_signals(ctypes.c_void_p(Xs.data_ptr()), engine._ld(Xs), int(Xs.shape[1]),
                                           int(Xs.shape[0]), float(scale), int(center), int(norm),
                                           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                   "lys_preproc_signals")
        return Xs
    if name == 'global_centering':                                   # preproc.py:55-57
        mean, _ = _feature_mean_std(Xs)
        return _feature_affine(Xs, mean, np.ones_like(mean))
    if name == 'global_standarization':                              # preproc.py:58-62 (sic)
        mean, std = _feature_mean_std(Xs)
        with np.errstate(divide='ignore'):
            return _feature_affine(Xs, mean, 1.0 / std)
    if name == 'whitening':                                          # preproc.py:77-78
        return zca_whiten_device(Xs)
    return Xs                                                        # unknown names pass through like the reference


class l2_normalizer():
    """lyssa/feature_extract/preproc.py:8-16 (host helper used on pooled cells)."""

    def __call__(self, Z):
        if Z.ndim == 1:
            return _normalize(Z)
        shape = Z.shape
        return _normalize(Z.flatten()).reshape(shape)


class preproc():
    """lyssa/feature_extract/preproc.py:46-80: X (n_features, n_samples) host array -> new float64 array."""

    def __init__(self, name):
        self.name = name

    def __call__(self, X):
        Xs = engine.signals_to_device(X)
        Xs = preproc_device(Xs, self.name)
        return Xs.t().contiguous().double().cpu().numpy()
