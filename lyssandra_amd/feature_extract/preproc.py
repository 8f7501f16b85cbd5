"""Per-patch preprocessing on the device -- lyssa/feature_extract/preproc.py:46-80.

Implemented: 'scaling', 'local_centering', 'contrast_normalization', 'normalization' (the per-datapoint ones that sit
directly in front of `sparse_encoder.encode`).  The per-FEATURE statistics ('global_centering',
'global_standarization') and ZCA 'whitening' are dataset-level passes outside the path and raise NotImplementedError.
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


def preproc_device(Xs, name):
    """In place on a signal-major fp32 cuda tensor [N, n]."""
    if name not in _FLAGS:
        raise NotImplementedError("preproc(%r) is not on the accelerated path" % (name,))
    torch = engine.require_gpu()
    lib = _lib.load()
    scale, center, norm = _FLAGS[name]
    _lib.check(lib.lys_preproc_signals(ctypes.c_void_p(Xs.data_ptr()), engine._ld(Xs), int(Xs.shape[1]), int(Xs.shape[0]),
                                       float(scale), int(center), int(norm),
                                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "lys_preproc_signals")
    return Xs


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
        preproc_device(Xs, self.name)
        return Xs.t().contiguous().double().cpu().numpy()
