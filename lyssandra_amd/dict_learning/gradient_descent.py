"""Drop-in for lyssa/dict_learning/gradient_descent.py (projected gradient descent dictionary learning).

Outside the kernels' headline path but it is the learner the reference's only
dictionary-learning test drives (lyssa/dict_learning/tests/test_dictionary_learn.py:11-21), so the drop-in
`sparse_encoder` has to work under it.  Per mini-batch: encode, grad = (DZ - X)Z' formed from the sparse codes as
D (ZZ') - XZ' (the online-DL statistics kernels), D <- norm_cols(clip(D - eta*grad + 2 mu D (D'D - I))) in one MFMA GEMM.
"""
import ctypes

import numpy as np

from .. import _lib, engine
from ..sparse_coding import sparse_encoder
from ..utils import gen_batches


def projected_grad_desc(X, n_atoms=None, sparse_coder=None, batch_size=None, D_init=None,
                        eta=None, mu=None, n_epochs=None, non_neg=False, verbose=False, n_jobs=1, mmap=False):
    """lyssa/dict_learning/gradient_descent.py:18-125.  Returns D (n_features, n_atoms) float64.

    The reference re-encodes the WHOLE data set after every mini-batch (:100) and discards the result; that call has
    no observable effect and is not reproduced.  The incoherence gradient is ADDED, as in the reference (:92).
    """
    if eta is None:
        raise ValueError('Must specify learning rate.')
    torch = engine.require_gpu()
    lib = _lib.load()
    sparse_coder.verbose = False
    X = np.asarray(X)
    n_features, n_samples = X.shape
    if D_init is None:
        from .utils import init_dictionary
        D, _ = init_dictionary(X, n_atoms, method='data', return_unused_data=True)
    else:
        D = D_init
    Xs = engine.signals_to_device(X)
    dd = engine.DeviceDictionary.from_host(D)
    device_coder = isinstance(sparse_coder, sparse_encoder) and sparse_coder.algorithm in ('bomp', 'omp', 'thresh')
    state = engine.OdlState(dd)                 # reuses its dA / dB buffers
    scratch = torch.empty((dd.Kp * dd.Kp + 2 * dd.Kp * dd.ldd,), dtype=torch.float32, device=dd.device)
    batch_idx = gen_batches(n_samples, batch_size=batch_size)
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)  # noqa: E731
    use_mu = mu is not None and mu > 0

    def encode(batch):
        Xb = Xs[batch.start:batch.stop]
        if device_coder:
            return Xb, sparse_coder.encode_device(Xb, dd)
        return Xb, engine.sparsify_host(sparse_coder(X[:, batch], dd.to_host()))

    max_patience = 10
    error_prev = 0
    patience = 0
    n_epochs = 1 if n_epochs is None else n_epochs
    for e in range(n_epochs):
        for batch in batch_idx:
            Xb, (idx, coef, nnz) = encode(batch)
            state._batch = (Xb, idx, coef, nnz)
            dA, dB = state.increments()
            G = dd.gram() if use_mu else None
            _lib.check(lib.lys_pgd_update(P(dd.D), P(dA), P(dB), P(G), dd.n, dd.K, float(eta),
                                          float(mu) if use_mu else 0.0, int(bool(non_neg)), P(scratch),
                                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "lys_pgd_update")
            dd.invalidate()
        if e < n_epochs - 1:
            error_curr = 0
            for batch in batch_idx:
                Xb, (idx, coef, nnz) = encode(batch)
                error_curr += engine.approx_error(Xb, dd, idx, coef, nnz)
            if verbose:
                print("end of epoch %d: error %.6g (diff %.6g)" % (e, error_curr, error_curr - error_prev))
                error_prev = error_curr
            if (e > 0) and (error_curr > 0.9 * error_prev or error_curr > error_prev):
                patience += 1
            if patience >= max_patience:
                break
    return dd.to_host()


class dictionary_learner():
    """lyssa/dict_learning/gradient_descent.py:128-160."""

    def __init__(self, n_atoms=None, sparse_coder=None, batch_size=None, eta=None, mu=None, D_init=None,
                 n_epochs=1, verbose=False, memory="low", mmap=False, non_neg=False, n_jobs=1):
        self.n_atoms = n_atoms
        self.sparse_coder = sparse_coder
        self.batch_size = batch_size
        self.eta = eta
        self.mu = mu
        self.n_epochs = n_epochs
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
        self.D = projected_grad_desc(X, n_atoms=self.n_atoms, sparse_coder=self.sparse_coder,
                                     batch_size=self.batch_size, mu=self.mu, D_init=self.D_init,
                                     eta=self.eta, n_epochs=self.n_epochs, verbose=self.verbose, n_jobs=self.n_jobs,
                                     non_neg=self.non_neg, mmap=self.mmap)

    def encode(self, X):
        Z = self.sparse_coder(X, self.D)
        return Z
