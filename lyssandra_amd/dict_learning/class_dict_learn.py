"""Class-specific K-SVD dictionaries (lyssa/dict_learning/class_dict_learn.py), the optional initialiser of LC-KSVD.

One `ksvd_dict_learn` per class on the class's columns; the class dictionaries are laid side by side.

Deviation, on purpose: the reference's `class_dict_learn` returns from INSIDE its class loop (class_dict_learn.py:124-139
-- the final `return D` is indented into the `for`), so it only ever learns class 0 and leaves the other classes' atoms
zero, which `lc_ksvd` then divides by (NaN dictionary).  Here every class is learned; `first_class_only=True` reproduces
the reference's early return.  The structural-incoherence option (`alpha`) imports a module that does not exist in the
reference (`lyssa.dict_learn`) and is not offered.
"""
import numpy as np

from .ksvd import ksvd_dict_learn


def class_dict_learn(X, y, n_class_atoms=None, sparse_coders=None, init_dict='data', max_iter=5, approx=False,
                     non_neg=False, eta=None, alpha=None, n_cycles=1, n_jobs=1, mmap=False, verbose=True,
                     first_class_only=False):
    if alpha is not None:
        raise NotImplementedError("structural incoherence (alpha) is broken in the reference too (class_dict_learn.py:128)")
    y = np.asarray(y)
    n_classes = len(set(y.tolist()))
    D = np.zeros((X.shape[0], int(np.sum(n_class_atoms))))
    for c in range(n_classes):
        Dc, _ = ksvd_dict_learn(X[:, y == c], n_class_atoms[c], init_dict='data', sparse_coder=sparse_coders[c],
                                max_iter=max_iter, non_neg=non_neg, approx=approx, eta=eta, n_cycles=n_cycles,
                                n_jobs=n_jobs, mmap=mmap, verbose=verbose)
        base = c * n_class_atoms[c]                      # :121 -- equal class sizes assumed, like the reference
        D[:, base:base + n_class_atoms[c]] = Dc
        if first_class_only:
            break
    return D


class class_ksvd_coder(object):
    """lyssa/dict_learning/class_dict_learn.py:16-96: keyword holder; `coder(X, y)` returns the joint dictionary."""

    def __init__(self, n_class_atoms=None, n_nonzero_coefs=None, atom_ratio=None, coef_ratio=None, sparse_coder=None,
                 non_neg=False, max_iter=None, approx=False, eta=None, alpha=None, n_cycles=1, n_jobs=1, mmap=False,
                 verbose=True):
        for name, value in list(locals().items()):
            if name != "self":
                setattr(self, name, value)
        self.D = None

    def _fit(self, X, y):
        y = np.asarray(y)
        n_classes = len(set(y.tolist()))
        if self.n_class_atoms is None:
            self.n_class_atoms = [int(np.sum(y == c) * self.atom_ratio) for c in range(n_classes)]
        if self.n_nonzero_coefs is None and self.coef_ratio is not None:
            self.n_nonzero_coefs = [int(self.n_class_atoms[c] * self.coef_ratio) for c in range(n_classes)]
        if not isinstance(self.n_class_atoms, list):
            self.n_class_atoms = [self.n_class_atoms] * n_classes
        self.D = class_dict_learn(X, y, n_class_atoms=self.n_class_atoms, sparse_coders=[self.sparse_coder] * n_classes,
                                  init_dict='data', max_iter=self.max_iter, non_neg=self.non_neg, approx=self.approx,
                                  eta=self.eta, alpha=self.alpha, n_cycles=self.n_cycles, n_jobs=self.n_jobs,
                                  mmap=self.mmap, verbose=self.verbose)

    def __call__(self, X, y):
        self._fit(X, y)
        return self.D

    def fit(self, X, y):
        self._fit(X, y)

    def encode(self, X):
        return self.sparse_coder.encode(X, self.D)
