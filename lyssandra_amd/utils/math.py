"""Host-side math shims with the reference's exact eps conventions (lyssa/utils/math.py).

These are tiny float64 numpy helpers used by HOST control flow only (e.g. normalising one replaced atom);
the hot path never goes through them.
"""
import numpy as np

_EPS = np.finfo(float).eps


def fast_dot(a, b):
    """lyssa/utils/math.py:11-24."""
    return np.dot(a, b)


def outer(a, b):
    return np.outer(a, b)


def norm(x):
    """lyssa/utils/math.py:52-54 (BLAS nrm2)."""
    x = np.asarray(x, dtype=np.float64)
    return float(np.sqrt(np.dot(x, x)))


def frobenius_squared(A):
    """lyssa/utils/math.py:57-58."""
    return np.sum(np.power(A, 2))


def normalize(x, eps=_EPS):
    """lyssa/utils/math.py:61-62 -- x / (||x|| + eps)."""
    return x / (norm(x) + eps)


def norm_cols(X, eps=_EPS):
    """lyssa/utils/math.py:65-71 -- in-place column normalisation, norms + eps."""
    norms = np.sqrt(np.einsum('ij,ij->j', X, X)) + eps
    X /= norms[np.newaxis, :]
    return X
