"""Drop-in for lyssa/dict_learning/gradient_descent.py (projected gradient descent dictionary learning).

Outside the kernels' headline path but it is the learner the reference's only
dictionary-learning test drives (lyssa/dict_learning/tests/test_dictionary_learn.py:11-21), so the drop-in
`sparse_encoder` has to work under it.  Per mini-batch: encode, grad = (DZ - X)Z' formed from the sparse codes as
D (ZZ') - XZ' (the online-DL statistics kernels), D <- norm_cols(clip(D - eta*grad + 2 mu D (D'D - I))) in one MFMA GEMM.
"""
import ctypes
from ._base import learner_shell, reference_patience, starting_dictionary

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
    X = np.asarray(X)
    setattr(sparse_coder, "verbose", False)
    D = starting_dictionary(X, n_atoms, D_init)
    Xs = engine.signals_to_device(X)
    dd = engine.DeviceDictionary.from_host(D)
    device_coder = isinstance(sparse_coder, sparse_encoder) and sparse_coder.algorithm in ('bomp', 'omp', 'thresh')
    state = engine.OdlState(dd)                 # reuses its dA / dB buffers
    scratch = torch.empty((dd.Kp * dd.Kp + 2 * dd.Kp * dd.ldd,), dtype=torch.float32, device=dd.device)
    batch_idx = gen_batches(X.shape[1], batch_size=batch_size)
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)  # noqa: E731
    use_mu = mu is not None and mu > 0

    def encode(batch):
        Xb = Xs[batch.start:batch.stop]
        if device_coder:
            return Xb, sparse_coder.encode_device(Xb, dd)
        return Xb, engine.sparsify_host(sparse_coder(X[:, batch], dd.to_host()))

    stop = reference_patience(verbose)
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
        if e == n_epochs - 1:
            break
        error = 0
        for batch in batch_idx:
            Xb, (idx, coef, nnz) = encode(batch)
            error += engine.approx_error(Xb, dd, idx, coef, nnz)
        if verbose:
            print("end of epoch %d: error %.6g (diff %.6g)" % (e, error, stop.change(error)))
        stop.observe(e, error)
        if stop.exhausted:
            break
    return dd.to_host()


class dictionary_learner(learner_shell):
    """lyssa/dict_learning/gradient_descent.py:128-160: keyword holder around ``projected_grad_desc``."""
    _forward = ("sparse_coder", "batch_size", "mu", "D_init", "eta", "n_epochs", "verbose", "n_jobs", "non_neg", "mmap")

    def __init__(self, n_atoms=None, sparse_coder=None, batch_size=None, eta=None, mu=None, D_init=None,
                 n_epochs=1, verbose=False, memory="low", mmap=False, non_neg=False, n_jobs=1):
        self._hold(locals())

    def _learn(self, X):
        self.D = projected_grad_desc(X, n_atoms=self.n_atoms, **self._learner_kwargs())
