"""Drop-in for lyssa/dict_learning/online_dict_learn.py (Mairal's online dictionary learning) on MI355X.

Host control flow of the reference (batching, beta schedule, epoch-end error pass, patience quirk) with the
per-batch arithmetic on the device: sparse coding, A += ZZ', B += XZ' from the sparse codes, the block
dictionary update D <- norm_cols(D + (B - DA) diag(1/(A_kk+eps))) with DA computed once per batch.
"""
import numpy as np

from .. import engine
from ..sparse_coding import sparse_encoder
from ..utils import gen_batches


def _is_device_coder(sc):
    return isinstance(sc, sparse_encoder) and sc.algorithm in ('bomp', 'omp', 'thresh', 'lasso')


def online_dict_learn(X, n_atoms, sparse_coder=None, batch_size=None, A=None, B=None, D_init=None,
                      beta=None, n_epochs=1, verbose=False, n_jobs=1, non_neg=False, mmap=False, group=None):
    """lyssa/dict_learning/online_dict_learn.py:18-124.  Returns ``(D, A, B)`` as float64 host arrays.

    Kept on purpose: ``beta=None`` => ``linspace(0, 1, n_iter)`` restarted at every epoch (:65-67,79), so the
    first batch of each epoch wipes A and B; ``DA`` is frozen per batch (:91); the epoch-end error pass and
    the patience quirk (:101-118).  ``D_init`` is updated in place like the reference (D = D_init, :46).
    ``group``: torch.distributed group when every rank holds a shard of each mini-batch (A|B all-reduced).
    """
    sparse_coder.verbose = False
    X = np.asarray(X)
    n_features, n_samples = X.shape
    if D_init is None:
        from .utils import init_dictionary
        D, unused_data = init_dictionary(X, n_atoms, method='data', return_unused_data=True)
    else:
        D = D_init
    Xs = engine.signals_to_device(X)
    dd = engine.DeviceDictionary.from_host(D)
    device_coder = _is_device_coder(sparse_coder)

    batch_idx = gen_batches(n_samples, batch_size=batch_size)
    n_batches = len(batch_idx)
    n_iter = n_batches
    if A is None and B is None:
        state = engine.OdlState(dd)
    else:
        state = engine.OdlState(dd, A=A, B=B)
    if beta is None:
        beta = np.linspace(0, 1, num=n_iter)
    else:
        beta = np.zeros(n_iter) + beta

    def encode(batch):
        Xb = Xs[batch.start:batch.stop]
        if device_coder:
            return Xb, sparse_coder.encode_device(Xb, dd)
        Zb = sparse_coder(X[:, batch], dd.to_host())
        return Xb, engine.sparsify_host(Zb)

    def finish():
        Dh = dd.to_host()
        if isinstance(D, np.ndarray) and D.shape == Dh.shape:
            D[:] = Dh  # in-place like the reference (D aliases D_init)
            return D, state.A_host(), state.B_host()
        return Dh, state.A_host(), state.B_host()

    max_patience = 10
    error_curr = 0
    error_prev = 0
    patience = 0
    for e in range(n_epochs):
        for i, batch in zip(range(n_iter), batch_idx):
            Xb, (idx, coef, nnz) = encode(batch)
            state.batch_update(Xb, idx, coef, nnz, beta[i], non_neg=non_neg, group=group)
        if e < n_epochs - 1:
            if patience >= max_patience:
                return finish()
            error_curr = 0
            for i, batch in zip(range(n_iter), batch_idx):
                Xb, (idx, coef, nnz) = encode(batch)
                error_curr += engine.approx_error(Xb, dd, idx, coef, nnz)
            if group is not None:  # every rank must take the same patience / early-return decisions
                import torch
                from .. import dist as _d
                t = torch.tensor([error_curr], dtype=torch.float64)
                _d.allreduce_sum_(t, group)
                error_curr = float(t.item())
            if verbose:
                print("end of epoch %d: error %.6g (diff %.6g)" % (e, error_curr, error_curr - error_prev))
                error_prev = error_curr
            if (e > 0) and (error_curr > 0.9 * error_prev or error_curr > error_prev):
                patience += 1
    return finish()


class online_dictionary_coder():
    """lyssa/dict_learning/online_dict_learn.py:127-160 -- keeps .D, .A, .B; A and B warm-start the next fit."""

    def __init__(self, n_atoms=None, sparse_coder=None, batch_size=None, beta=None, D_init=None,
                 n_epochs=1, verbose=False, memory="low", mmap=False, non_neg=False, n_jobs=1):
        self.n_atoms = n_atoms
        self.sparse_coder = sparse_coder
        self.batch_size = batch_size
        self.beta = beta
        self.n_epochs = n_epochs
        self.A = None
        self.B = None
        self.D_init = D_init
        self.memory = memory
        self.verbose = verbose
        self.n_jobs = n_jobs
        self.mmap = mmap
        self.non_neg = non_neg

    def __call__(self, X):
        self.fit(X)
        return self.encode(X)

    def fit(self, X):
        self.D, self.A, self.B = online_dict_learn(X, self.n_atoms, sparse_coder=self.sparse_coder,
                                                   batch_size=self.batch_size, A=self.A, B=self.B, D_init=self.D_init,
                                                   beta=self.beta, n_epochs=self.n_epochs, verbose=self.verbose,
                                                   n_jobs=self.n_jobs, non_neg=self.non_neg, mmap=self.mmap)

    def encode(self, X):
        Z = self.sparse_coder(X, self.D)
        return Z
