"""Drop-in for lyssa/dict_learning/ksvd.py (approximate K-SVD path) on MI355X.

``approx_ksvd``, ``ksvd_dict_learn`` and ``ksvd_coder`` keep the reference's signatures, in-place mutation
and host control flow (patience quirk, unused-atom replacement with the GLOBAL numpy RNG); the arithmetic --
encode, residual, per-atom update, error -- runs in liblyssa_hip.so on device-resident data: X is uploaded
once, the codes stay as a sparse triplet on the device for the whole fit.
"""
import time

import numpy as np

from .. import engine
from ..sparse_coding import sparse_encoder
from ..utils.math import normalize
from ._base import learner_shell, reference_patience


def _is_device_coder(sc):
    return isinstance(sc, sparse_encoder) and sc.algorithm in ('bomp', 'omp', 'thresh', 'lasso')


def approx_ksvd(Y, D, X, n_cycles=1, verbose=True):
    """lyssa/dict_learning/ksvd.py:98-126.  Y (n, N) data, D (n, K), X (K, N) dense codes.

    Mutates D and X IN PLACE like the reference and returns ``(D, X, unused_atoms)``.  The support pattern
    of X is preserved; atoms are visited in order 0..K-1 (Gauss-Seidel through the shared residual).
    """
    Ys = engine.signals_to_device(Y)
    dd = engine.DeviceDictionary.from_host(D)
    idx, coef, nnz = engine.sparsify_host(X)
    R, _ = engine.residual(Ys, dd, idx, coef, nnz, want_R=True, want_err=False)
    unused_atoms = []
    for _ in range(n_cycles):
        unused_atoms += engine.ksvd_cycle(R, dd, idx, coef, nnz)
    D[:] = dd.to_host()
    _scatter_codes(X, idx, coef, nnz)
    return D, X, unused_atoms


def ksvd(Y, D, X, n_cycles=1, verbose=True, group=None):
    """lyssa/dict_learning/ksvd.py:19-43 -- the exact K-SVD update (rank-1 SVD of every atom's restricted residual).

    Same in-place contract as ``approx_ksvd``.  The reference's ``randomized_svd(n_iter=10, flip_sign=False)`` is
    replaced by a deterministic device eigen-solve (Gram matrix + Lanczos started from the old atom), so atoms agree with the
    reference up to the sign of (d_k, x_k) and the accuracy of the randomized solver; D X is sign-invariant.
    ``group`` (extension): a torch.distributed process group -- Y and X then hold THIS rank's columns (signal shard), D is
    replicated; per atom the ranks exchange the n x n Gram matrix (n <= 256) or one n-vector per power iteration (n > 256).
    """
    Ys = engine.signals_to_device(Y)
    dd = engine.DeviceDictionary.from_host(D)
    idx, coef, nnz = engine.sparsify_host(X)
    R, _ = engine.residual(Ys, dd, idx, coef, nnz, want_R=True, want_err=False)
    unused_atoms = []
    buffers = {}
    for _ in range(n_cycles):
        unused_atoms += engine.ksvd_exact_cycle(R, dd, idx, coef, nnz, buffers=buffers, group=group)
    D[:] = dd.to_host()
    _scatter_codes(X, idx, coef, nnz)
    return D, X, unused_atoms


def nn_ksvd(Y, D, X, n_cycles=1, verbose=True):
    """lyssa/dict_learning/ksvd.py:46-95 -- the non-negative K-SVD update: rank-1 solve per atom, projection of (d, x) onto
    the non-negative orthant, ``n_cycles`` alternating projections, renormalisation.  One pass over the atoms; an atom whose
    projected d or x vanishes keeps its column, codes and residual (:79-82).  Same in-place contract as ``approx_ksvd``.
    Sign of the rank-1 pair: u . d_old >= 0 (the reference's randomized_svd leaves it to chance)."""
    Ys = engine.signals_to_device(Y)
    dd = engine.DeviceDictionary.from_host(D)
    idx, coef, nnz = engine.sparsify_host(X)
    R, _ = engine.residual(Ys, dd, idx, coef, nnz, want_R=True, want_err=False)
    unused_atoms = engine.ksvd_exact_cycle(R, dd, idx, coef, nnz, nn_cycles=n_cycles)
    D[:] = dd.to_host()
    _scatter_codes(X, idx, coef, nnz)
    return D, X, unused_atoms


def _scatter_codes(X, idx, coef, nnz):
    """Write the (updated) coefficients back into the dense host matrix, support unchanged."""
    hi, hc, hn = idx.cpu().numpy(), coef.cpu().numpy(), nnz.cpu().numpy()
    N, k = hi.shape
    valid = np.arange(k)[None, :] < hn[:, None]
    cols = np.broadcast_to(np.arange(N)[:, None], (N, k))
    X[hi[valid], cols[valid]] = hc[valid]


def ksvd_dict_learn(X, n_atoms, init_dict='data', sparse_coder=None,
                    max_iter=20, non_neg=False, approx=False, eta=None,
                    n_cycles=1, n_jobs=1, mmap=False, verbose=True, return_codes=True, group=None, shard_span=None,
                    n_total=None):
    """lyssa/dict_learning/ksvd.py:129-231 (``approx=True``: approximate update; ``approx=False``: exact rank-1 update).

    Returns ``(D, Z)`` with D float64 (n, K) and Z dense float64 (K, N) -- pass ``return_codes=False`` to skip
    the dense materialisation (then Z is the device triplet).  Reference behaviours kept on purpose:
      * patience: ``error_prev`` is refreshed only when ``verbose`` and *before* the test (:222-228), so the
        loop stops after 11 iterations whatever ``max_iter`` is;
      * unused atoms are replaced by random unused datapoints drawn with ``np.random.choice`` (:199-207);
      * the error is evaluated with the atom-updated codes and the final dictionary (:220).
    ``group``: torch.distributed group when X is THIS RANK'S SHARD of the signals, columns
    ``shard_span = (start, stop)`` of ``n_total`` (lyssandra_amd.dist.local_shard): per-atom statistics and the error
    are all-reduced, `init_dict='data'` and the unused-atom replacement work on GLOBAL signal indices (every rank
    must hold the same numpy RNG state).  The returned codes are the local shard's.
    """
    if max_iter is None:
        # the reference's default (ksvd_coder(max_iter=None)): under Python 2 `0 < None` is False, the loop never runs and
        # the initial dictionary is returned with all-zero codes -- reproduced explicitly instead of a py3 TypeError
        max_iter = 0
    X = np.asarray(X)
    n_samples = X.shape[1]
    unused_data = []
    if group is not None:
        from .. import dist as _dist
        if shard_span is None or n_total is None:
            raise ValueError("group mode needs shard_span=(start, stop) and n_total")
    if isinstance(init_dict, str) and init_dict == 'data':
        if group is None:
            from .utils import init_dictionary
            D, unused_data = init_dictionary(X, n_atoms, method=init_dict, return_unused_data=True)
        else:
            D, unused_data = _dist.init_dictionary_sharded(X, shard_span, n_total, n_atoms, group)
    else:
        D = np.copy(init_dict)
    Xs = engine.signals_to_device(X)
    dd = engine.DeviceDictionary.from_host(D)
    device_coder = _is_device_coder(sparse_coder)
    if mmap and sparse_coder is not None:
        sparse_coder.mmap = True
    if verbose:
        print("dictionary initialized")
    stop = reference_patience(verbose)
    it = 0
    idx = coef = nnz = None
    out = None            # code triplet, residual and index buffers are allocated once and reused every iteration
    R = None
    buffers = {}
    while it < max_iter and not stop.exhausted:
        it_start = time.time()
        # ---- sparse coding
        if device_coder:
            idx, coef, nnz = out = sparse_coder.encode_device(Xs, dd, out=out)
        else:
            idx, coef, nnz = engine.sparsify_host(sparse_coder(X, dd.to_host()))
        t_sparse = time.time() - it_start
        # ---- approximate K-SVD atom updates
        R, _ = engine.residual(Xs, dd, idx, coef, nnz, want_R=True, want_err=False,
                               out=R if (R is not None and R.shape[0] == idx.shape[0]) else None)
        unused_atoms = []
        if non_neg and not approx:
            # ksvd.py:187-188: nn_ksvd with n_cycles = the ITERATION INDEX (alternating projections), one pass over the atoms
            unused_atoms += engine.ksvd_exact_cycle(R, dd, idx, coef, nnz, buffers=buffers, group=group, nn_cycles=it)
        for _ in range(0 if (non_neg and not approx) else n_cycles):
            if approx:
                unused_atoms += engine.ksvd_cycle(R, dd, idx, coef, nnz, group=group, buffers=buffers)
            else:  # ksvd.py:189-190: exact rank-1 update
                unused_atoms += engine.ksvd_exact_cycle(R, dd, idx, coef, nnz, buffers=buffers, group=group)
        # ---- replace unused atoms (host RNG, ksvd.py:199-207)
        for atom in unused_atoms:
            if not unused_data:
                break
            pick = np.random.choice(unused_data, size=1)[0]        # one draw per replaced atom, global RNG
            col = X[:, pick] if group is None else _dist.fetch_global_column(X, shard_span, int(pick), group)
            dd.set_atom(atom, normalize(np.asarray(col, dtype=np.float64)))
            unused_data.remove(pick)
        # ---- force mutual incoherence, not in the last iteration (ksvd.py:209-213)
        if eta is not None and it < max_iter - 1:
            Dh = dd.to_host()
            if group is None:
                from .utils import force_mi
                Dh, unused_data = force_mi(Dh, X, (idx, coef, nnz), unused_data, eta)
            else:   # replicated decision: code-row norms all-reduced, candidate columns fetched by global index
                Dh, unused_data = _dist.force_mi_sharded(Dh, X, (idx, coef, nnz), shard_span, unused_data, eta, group)
            dd.set(Dh)
        # ---- error with the updated codes (ksvd.py:220).  The single-GPU block sweep's final pass leaves ||R||^2 = ||X - D Z||^2
        # behind (engine.sweep_error); replaced UNUSED atoms have all-zero code rows and do not change it, `eta` does
        error = engine.sweep_error(buffers) if (group is None and not (eta is not None and it < max_iter - 1)) else None
        if error is None:
            error = engine.approx_error(Xs, dd, idx, coef, nnz)
        if group is not None:
            error = _allreduce_scalar(error, dd.device, group)
        if verbose:
            print("iteration %d: sparse coding %.3fs, total %.3fs, unused atoms %d, error %.6g (diff %.6g)"
                  % (it, t_sparse, time.time() - it_start, len(unused_atoms), error, stop.change(error)))
        stop.observe(it, error)
        it += 1
    D_out = dd.to_host()
    if idx is None:
        Z = np.zeros((n_atoms, n_samples))
    elif return_codes:
        Z = engine.densify(idx, coef, nnz, n_atoms)
    else:
        Z = (idx, coef, nnz)
    return D_out, Z


def _allreduce_scalar(v, device, group):
    import torch
    import torch.distributed as dist
    t = torch.tensor([v], dtype=torch.float64, device=device)
    dist.all_reduce(t, group=group)
    return float(t.item())


class ksvd_coder(learner_shell):
    """lyssa/dict_learning/ksvd.py:234-271: keyword holder around ``ksvd_dict_learn``; the dictionary ends up in ``.D``.
    ``n_nonzero_coefs`` is accepted and ignored like in the reference (the sparsity lives in ``sparse_coder.params``)."""
    _forward = ("init_dict", "sparse_coder", "max_iter", "non_neg", "approx", "eta", "n_cycles", "n_jobs", "mmap",
                "verbose")

    def __init__(self, n_atoms=None, n_nonzero_coefs=None, sparse_coder=None, init_dict="data",
                 max_iter=None, non_neg=False, approx=True, eta=None, n_cycles=1, n_jobs=1,
                 mmap=False, verbose=True):
        self._hold(locals())

    def _learn(self, X):
        self.D, _ = ksvd_dict_learn(X, self.n_atoms, return_codes=False, **self._learner_kwargs())
