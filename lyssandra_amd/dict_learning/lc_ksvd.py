"""Drop-in for lyssa/dict_learning/lc_ksvd.py (Label-Consistent K-SVD, Jiang et al.) on MI355X -- SURVEY 8f rank 4.

LC-KSVD learns the dictionary D, a linear transform G of the codes towards the "discriminative" target Q and a linear
classifier W jointly, as ONE exact K-SVD on stacked data (lc_ksvd.py:140-188):

    [ X ; sqrt(alpha) Q ; sqrt(beta) H ]  ~  [ D ; sqrt(alpha) G ; sqrt(beta) W ] Z

The codes come from the sparse coder on the top block only (:165), the stacked dictionary is updated by the exact
rank-1 `ksvd` (:172; `approx` is accepted and ignored by the reference), then every stacked atom is rescaled by the
norm of its D part (:180-188).  Device work: the encode (`sparse_encoder`) and the exact K-SVD sweep -- the stack has
n_features + n_atoms + n_classes rows, i.e. the "tall" shape of `lys_ksvd_exact_sweep` (few signals per atom, many
features).  The O(nK) stacking / rescaling between them stays on the host in float64 like the reference.
"""
import numpy as np

from ..classify import classifier
from .ksvd import ksvd
from .utils import init_dictionary, approx_error


def _label_matrix(y, n_classes):
    """H (n_classes, n_samples): H[c, i] = 1 iff sample i has label c (lc_ksvd.py:124-128)."""
    y = np.asarray(y).astype(int)
    H = np.zeros((n_classes, y.size))
    H[y, np.arange(y.size)] = 1
    return H


def _ridge(Z, T, lam):
    """(Z Z' + lam I)^-1 Z T', transposed -- the closed-form initialisations of W and G (lc_ksvd.py:136-139)."""
    A = Z @ Z.T + lam * np.eye(Z.shape[0])
    return np.linalg.solve(A, Z @ T.T).T


def _stack(D, G, W, alpha, beta):
    """Rescale every atom (and its G / W columns) by the norm of its D part, then stack (lc_ksvd.py:148-156,180-188)."""
    scale = np.sqrt(np.einsum('ij,ij->j', D, D))
    with np.errstate(divide='ignore', invalid='ignore'):
        D, G, W = D / scale, G / scale, W / scale
    return D, G, W, np.vstack((D, np.sqrt(alpha) * G, np.sqrt(beta) * W))


def lc_ksvd(X, y, D, Q, alpha=1, beta=1, lambda1=1, lambda2=1,
            sparse_coder=None, max_iter=2, approx=False, mmap=False, verbose=False, n_jobs=1, group=None):
    """lyssa/dict_learning/lc_ksvd.py:105-216.  X (n_features, n_samples), y labels 0..C-1, D (n_features, n_atoms)
    initial dictionary, Q (n_atoms, n_samples) with Q[k, i] = 1 iff atom k and sample i share a class.
    Returns ``(D, Z, W)``: Z are the last iteration's codes AFTER the K-SVD coefficient update (the reference hands its
    Z to `ksvd`, which updates it in place).
    ``group`` (extension, round 4): a torch.distributed process group -- X, y, Q then hold THIS rank's samples, D is
    replicated; the stacked exact K-SVD runs on signal shards (n > 256: one n-vector all-reduce per power iteration,
    dist.ksvd_exact_cycle_sharded_mf), everything else of the loop is per sample or replicated.  Returns this rank's Z."""
    n_features, K = X.shape[0], D.shape[1]
    n_classes = len(set(np.asarray(y).tolist()))
    if group is not None:  # a shard need not hold every class: the labels are 0 .. C-1
        import torch
        from .. import dist as _ldist
        top = torch.tensor([int(np.max(np.asarray(y).astype(int))) + 1 if np.size(y) else 0], dtype=torch.int64)
        n_classes = int(_ldist.allreduce_max_(top, group=group).item())  # staged through the device under RCCL
    H = _label_matrix(y, n_classes)
    Z = np.zeros((K, X.shape[1]))
    # with Z = 0 these are zero matrices (:136-139) -- kept as the reference computes them
    W = _ridge(Z, H, lambda1)
    G = _ridge(Z, Q, lambda2)
    stacked_X = np.vstack((X, np.sqrt(alpha) * Q, np.sqrt(beta) * H))
    D, G, W, stacked_D = _stack(np.array(D, dtype=np.float64), G, W, alpha, beta)
    last_error = 0
    for it in range(max_iter):
        Z = sparse_coder(X, D)
        stacked_D, _, unused_atoms = ksvd(stacked_X, stacked_D, Z, verbose=False, group=group)
        if verbose:
            print("iteration %d: number of unused atoms: %d" % (it, len(unused_atoms)))
        D, G, W, stacked_D = _stack(stacked_D[:n_features], stacked_D[n_features:n_features + K],
                                    stacked_D[n_features + K:], alpha, beta)
        if verbose:
            error = approx_error(D, Z, X, n_jobs=2)
            acc = np.mean(np.argmax(W @ Z, axis=0) == np.asarray(y).astype(int))
            print("error: %g  error difference: %g  classification accuracy: %g" % (error, error - last_error, acc))
            last_error = error
    return D, Z, W


def lc_ksvd_predict(X, D, W, sparse_coder):
    """lyssa/dict_learning/lc_ksvd.py:92-102: label = argmax_c (W z)_c of every column's code."""
    Z = sparse_coder(X, D)
    return [int(c) for c in np.argmax(W @ Z, axis=0)]


class lc_ksvd_classifier(classifier):
    """lyssa/dict_learning/lc_ksvd.py:16-89.  `train` builds the initial dictionary (a `class_dict_coder`, or
    `n_class_samples` atoms per class drawn from the class's data with the global RNG), the target Q and runs `lc_ksvd`."""

    def __init__(self, class_dict_coder=None, param_grid=None,
                 sparse_coder=None, max_iter=2, approx=True, eta=0,
                 n_class_samples=None, n_test_samples=None, n_tests=1, n_folds=None,
                 alpha=1, beta=1, mmap=False, verbose=False, n_jobs=1):
        classifier.__init__(self, n_folds=n_folds, param_grid=param_grid, n_class_samples=n_class_samples,
                            n_test_samples=n_test_samples, n_tests=n_tests, name='lc_ksvd_classifier')
        kept = dict(class_dict_coder=class_dict_coder, sparse_coder=sparse_coder, max_iter=max_iter, approx=approx,
                    alpha=alpha, beta=beta, mmap=mmap, verbose=verbose, n_jobs=n_jobs)
        for name, value in kept.items():
            setattr(self, name, value)
        self.n_class_atoms = None          # per-class atom counts, fixed by the first `train`
        sparse_coder.n_jobs = n_jobs       # the reference pokes the coder (:45)

    def train(self, X_train, y_train, param_set=None):
        if param_set is not None:
            self.alpha, self.beta = param_set['alpha'], param_set['beta']
        y_train = np.asarray(y_train)
        classes = range(len(set(y_train.tolist())))
        if self.class_dict_coder is None:
            # one block of atoms per class, drawn from the class's own columns (global RNG, lc_ksvd.py:56-68)
            if self.n_class_atoms is None:
                self.n_class_atoms = np.full(len(classes), self.n_class_samples, dtype=int)
            blocks = [init_dictionary(X_train[:, y_train == c], self.n_class_atoms[c], method='data', normalize=True)
                      for c in classes]
            D = np.zeros((X_train.shape[0], int(np.sum(self.n_class_atoms))))
            for c, Dc in zip(classes, blocks):
                first = c * self.n_class_atoms[c]                      # (:66 -- equal class sizes assumed)
                D[:, first:first + Dc.shape[1]] = Dc
        else:
            D = self.class_dict_coder(X_train, y_train)
            self.n_class_atoms = self.class_dict_coder.n_class_atoms
        # Q[k, i] = 1 iff atom k and sample i belong to the same class (:73-80)
        owner = np.repeat(np.arange(len(classes)), np.asarray(self.n_class_atoms, dtype=int))
        Q = (owner[:, None] == y_train[None, :]).astype(np.float64)
        self.D, Z, self.W = lc_ksvd(X_train, y_train, D, Q, sparse_coder=self.sparse_coder, alpha=self.alpha,
                                    beta=self.beta, lambda1=1, lambda2=1, max_iter=self.max_iter, approx=self.approx,
                                    verbose=self.verbose, n_jobs=self.n_jobs)

    def predict(self, X_test):
        return lc_ksvd_predict(X_test, self.D, self.W, self.sparse_coder)
