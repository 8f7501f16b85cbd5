"""Dictionary-learning helpers (lyssa/dict_learning/utils.py), host control flow + device reductions."""
import numpy as np

from .. import engine
from ..utils.math import norm_cols, normalize as _normalize


def init_dictionary(X, n_atoms, method='data', return_unused_data=False, normalize=True):
    """lyssa/dict_learning/utils.py:35-75, methods 'data', 'random' and 'svd'.

    Host logic, global numpy RNG exactly like the reference (:60): candidates are the columns with energy
    > 1e-6, ``np.random.choice(len(idxs), n_atoms, replace=False)``, D = X[:, chosen] (a copy), optional
    norm_cols, ``unused_data`` = remaining candidate indices as a Python list.
    """
    X = np.asarray(X)
    if method == "data":
        energy = np.einsum('ij,ij->j', X, X)
        idxs = np.flatnonzero(energy > 1e-6)                 # candidates, ascending (kept as an array: N can be 10^6)
        if idxs.size < n_atoms:
            raise ValueError("not enough datapoints to initialize the dictionary")
        subset = np.random.choice(idxs.size, size=n_atoms, replace=False)      # the reference's draw (:60)
        D = np.array(X[:, idxs[subset]], dtype=np.float64)
        if normalize:
            D = norm_cols(D)
        if return_unused_data:
            keep = np.ones(idxs.size, dtype=bool)
            keep[subset] = False
            return D, idxs[keep].tolist()                      # remaining candidates, ascending, a Python list
        return D
    if method == "random":
        D = np.random.randn(X.shape[0], n_atoms)
        return norm_cols(D)
    if method == "svd":
        # :38-48: left singular vectors of X, zero-padded when n_atoms exceeds the rank (one-off host LAPACK call on an
        # n x N matrix, like the reference; the reference computes the matching codes too but returns only D)
        U, _, _ = np.linalg.svd(X, full_matrices=False)
        r = U.shape[1]
        if n_atoms <= r:
            return np.array(U[:, :n_atoms], dtype=np.float64)
        return np.c_[U, np.zeros((U.shape[0], n_atoms - r))]
    raise ValueError("init_dictionary: unknown method %r" % (method,))


def approx_error(D, Z, X, n_jobs=1):
    """lyssa/dict_learning/utils.py:14-19 -- ||X - DZ||_F^2, evaluated on the device from the sparse form of Z."""
    Xs = engine.signals_to_device(X)
    dd = engine.DeviceDictionary.from_host(D)
    idx, coef, nnz = engine.sparsify_host(Z)
    return engine.approx_error(Xs, dd, idx, coef, nnz)


def average_mutual_coherence(D):
    """lyssa/dict_learning/utils.py:7-11 -- mean off-diagonal |D'D| (Gram from the MFMA GEMM, reduction on the device)."""
    import ctypes
    from .. import _lib
    torch = engine.require_gpu()
    lib = _lib.load()
    D = np.asarray(D)
    dd = engine.DeviceDictionary.from_host(D)
    K = dd.K
    G = dd.gram()
    out = torch.zeros((1,), dtype=torch.float64, device=dd.device)
    _lib.check(lib.lys_offdiag_abs_sum(ctypes.c_void_p(G.data_ptr()), K, ctypes.c_void_p(out.data_ptr()),
                                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "lys_offdiag_abs_sum")
    return float(out.item()) / float(K * (K - 1))


def _code_row_norms(Z, n_atoms):
    """||Z[k, :]|| for every atom, from a dense host matrix or from the device triplet (idx, coef, nnz)."""
    if isinstance(Z, tuple):
        torch = engine.require_gpu()
        idx, coef, nnz = Z
        k = idx.shape[1]
        valid = torch.arange(k, device=idx.device)[None, :] < nnz[:, None]
        sq = torch.zeros((n_atoms,), dtype=torch.float64, device=idx.device)
        sq.index_add_(0, idx[valid].long(), coef[valid].double() ** 2)
        return np.sqrt(sq.cpu().numpy())
    Z = np.asarray(Z)
    return np.sqrt(np.einsum('ij,ij->i', Z, Z))


def force_mi(D, X, Z, unused_data, eta, max_tries=100, fetch_column=None, usage=None):
    """lyssa/dict_learning/utils.py:86-139 -- replace atoms whose mutual coherence exceeds ``eta`` by datapoints.

    Host control flow with the GLOBAL numpy RNG like the reference: the coherence matrix |D'D| is computed ONCE (:89)
    and not refreshed after a replacement; for atom i the most coherent partner j is looked up in that matrix, the one
    of the two with the SMALLER code-row norm is replaced (:100-103: ``norm(Z[i]) > norm(Z[j])`` selects i ... the
    reference's comment asks "the one least used?", its code picks the more used one -- reproduced as written); up to
    ``max_tries`` + 1 random unused datapoints are tried and the least coherent one is taken.  Returns
    ``(D, unused_data)``; D is modified in place.  Deviations, both on paths where the reference raises: when no
    candidate lowers the coherence (``min_idx is None`` at :134) the atom is left alone, and when the candidate list
    runs dry (:119-120, the reference returns a bare ``D`` that its caller cannot unpack) ``(D, unused_data)`` is
    returned.
    Signal shards (lyssandra_amd.dist.force_mi_sharded): ``fetch_column(i)`` returns GLOBAL column i on every rank and
    ``usage`` the code-row norms over all shards; D and the RNG state are replicated, so every rank takes the same decisions.
    """
    D = np.asarray(D)
    K = D.shape[1]
    coherence = np.abs(D.T @ D)            # computed once, never refreshed (:89)
    coherence[np.diag_indices(K)] = 0.0
    if usage is None:
        usage = _code_row_norms(Z, K)

    def column(i):
        raw = X[:, i] if fetch_column is None else fetch_column(int(i))
        return _normalize(np.asarray(raw, dtype=np.float64))

    for first in range(K):
        partner = int(coherence[first].argmax())
        worst = coherence[first, partner]
        if worst < eta:
            continue
        victim = first if usage[first] > usage[partner] else partner      # as written at :100-103
        pool = list(unused_data)
        best, best_coh, now = None, worst, worst
        # up to max_tries + 1 draws from the GLOBAL numpy RNG, one `np.random.choice(pool, size=1)` each (:117-133)
        for _ in range(max_tries + 1):
            if not now > eta:
                break
            if not pool:
                return D, unused_data
            pick = np.random.choice(pool, size=1)[0]
            pool.remove(pick)
            now = np.abs(D.T @ column(pick)).max()
            if now < best_coh:
                best, best_coh = pick, now
        if best is not None:
            D[:, victim] = column(best)
            unused_data.remove(best)
    return D, unused_data
