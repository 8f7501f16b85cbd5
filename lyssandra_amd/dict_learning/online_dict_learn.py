"""Drop-in for lyssa/dict_learning/online_dict_learn.py (Mairal's online dictionary learning) on MI355X.

Host control flow of the reference (batching, beta schedule, epoch-end error pass, patience quirk) with the
per-batch arithmetic on the device: sparse coding, A += ZZ', B += XZ' from the sparse codes, the block
dictionary update D <- norm_cols(D + (B - DA) diag(1/(A_kk+eps))) with DA computed once per batch.
"""
import numpy as np
from ._base import learner_shell, reference_patience, starting_dictionary

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
    X = np.asarray(X)
    setattr(sparse_coder, "verbose", False)                        # :41
    D = starting_dictionary(X, n_atoms, D_init)
    Xs = engine.signals_to_device(X)
    dd = engine.DeviceDictionary.from_host(D)
    device_coder = _is_device_coder(sparse_coder)
    # `batch_size` may also be an explicit list of column ranges (dist.shard_minibatches with uneven shards)
    batch_idx = list(batch_size) if isinstance(batch_size, (list, tuple)) else gen_batches(X.shape[1], batch_size=batch_size)
    n_iter = len(batch_idx)
    warm = not (A is None and B is None)                           # :59-63: both or none
    state = engine.OdlState(dd, A=A, B=B) if warm else engine.OdlState(dd)
    # forgetting factors of one epoch: a ramp 0 .. 1 (restarted every epoch, so its first batch wipes A and B), or constant
    beta = np.linspace(0, 1, num=n_iter) if beta is None else np.full(n_iter, beta, dtype=np.float64)

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

    stop = reference_patience(verbose)
    for e in range(n_epochs):
        for i, batch in zip(range(n_iter), batch_idx):
            Xb, (idx, coef, nnz) = encode(batch)
            state.batch_update(Xb, idx, coef, nnz, beta[i], non_neg=non_neg, group=group)
        if e == n_epochs - 1:
            break                                                  # no error pass after the last epoch (:101)
        if stop.exhausted:
            return finish()
        error = 0
        for i, batch in zip(range(n_iter), batch_idx):
            Xb, (idx, coef, nnz) = encode(batch)
            error += engine.approx_error(Xb, dd, idx, coef, nnz)
        if group is not None:  # every rank must take the same patience / early-return decisions
            import torch
            from .. import dist as _d
            t = torch.tensor([error], dtype=torch.float64)
            _d.allreduce_sum_(t, group)
            error = float(t.item())
        if verbose:
            print("end of epoch %d: error %.6g (diff %.6g)" % (e, error, stop.change(error)))
        stop.observe(e, error)
    return finish()


class online_dictionary_coder(learner_shell):
    """lyssa/dict_learning/online_dict_learn.py:127-160: keeps ``.D``, ``.A``, ``.B``; the statistics A and B warm-start
    the next ``fit`` while ``D_init`` is NOT refreshed (the reference's behaviour, pinned by the warm-start golden)."""
    _forward = ("sparse_coder", "batch_size", "D_init", "beta", "n_epochs", "verbose", "n_jobs", "non_neg", "mmap")

    def __init__(self, n_atoms=None, sparse_coder=None, batch_size=None, beta=None, D_init=None,
                 n_epochs=1, verbose=False, memory="low", mmap=False, non_neg=False, n_jobs=1):
        self._hold(locals())
        self.A = self.B = None

    def _learn(self, X):
        self.D, self.A, self.B = online_dict_learn(X, self.n_atoms, A=self.A, B=self.B, **self._learner_kwargs())
