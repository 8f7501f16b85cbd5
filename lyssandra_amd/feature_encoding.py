"""Drop-in for lyssa/feature_encoding.py (Coates-Ng encoders) on the same engine.

`soft_thresholding` (:26-37) is the reference's `thresholding` under another name: the k largest SIGNED
correlations of D'X.  `feature_encoder(algorithm='soft_thresholding')` (:40-89) therefore runs the alpha0 MFMA GEMM +
`thresh_wave_kernel`.  (`sign_splitting`, :14-23, indexes `np.where(...)[0]` as if it were a pair and does not run in
the reference; it is not reproduced.)
"""
import numpy as np

from . import engine
from .sparse_coding import sparse_encoder, _empty_mmap


def soft_thresholding(Alpha, nonzero_percentage=None, n_nonzero_coefs=None):
    """lyssa/feature_encoding.py:26-37 for a precomputed Alpha (n_atoms, n_samples) host array."""
    from .sparse_coding import _thresh_from_alpha
    return _thresh_from_alpha(Alpha, nonzero_percentage, n_nonzero_coefs)


class feature_encoder(object):
    """lyssa/feature_encoding.py:40-89."""

    def __init__(self, algorithm=None, params=None, n_jobs=1, verbose=True, mmap=False):
        settings = dict(algorithm=algorithm, params={} if params is None else params, n_jobs=n_jobs, verbose=verbose,
                        mmap=mmap)
        for name, value in settings.items():
            setattr(self, name, value)

    def encode(self, X, D):
        return self.__call__(X, D)

    def __call__(self, X, D):
        if self.algorithm != 'soft_thresholding':
            # the reference leaves `func` unbound for anything else and dies with a NameError (:82)
            raise ValueError("feature encoder %r not found" % (self.algorithm,))
        se = sparse_encoder(algorithm='thresh', params=self.params, n_jobs=self.n_jobs, verbose=self.verbose,
                            mmap=self.mmap)
        return se.encode(X, D)
