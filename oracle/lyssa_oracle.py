"""CPU oracle: our own float64 restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY.  This module is the *checker*: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it.
Nothing under ``lyssandra_amd/`` imports it; the product path is HIP-only and fails
loudly when the HIP library is missing.

Parity pinning: the reference's own tests hold **no** golden vector or known-answer test
for this path, so this restatement is pinned against outputs of the
reference itself: ``oracle/make_golden.py`` imports the (py3-converted) reference in the
build container, runs it on seeded inputs and commits inputs+outputs under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks every function below against
those vectors (identical supports, <=1e-12 on values).

Every function cites the reference lines it restates (paths relative to /root/reference).
Conventions are the reference's: X is (n_features, n_samples), D is (n_features, n_atoms),
Z is dense (n_atoms, n_samples), all float64.
"""
import numpy as np
from scipy.linalg import solve_triangular

EPS64 = float(np.finfo(np.float64).eps)


# --------------------------------------------------------------------------- math shims
def fast_dot(a, b):
    """lyssa/utils/math.py:11-24 -- matrix product (np.dot / ddot)."""
    return np.dot(a, b)


def norm(x):
    """lyssa/utils/math.py:52-54 -- BLAS nrm2."""
    return float(np.sqrt(np.dot(x, x)))


def frobenius_squared(A):
    """lyssa/utils/math.py:57-58."""
    return float(np.sum(np.power(A, 2)))


def normalize(x, eps=EPS64):
    """lyssa/utils/math.py:61-62 -- x / (||x|| + eps)."""
    return x / (norm(x) + eps)


def norm_cols(X, eps=EPS64):
    """lyssa/utils/math.py:65-71 -- in-place column normalisation with +eps."""
    norms = np.sqrt(np.einsum('ij,ij->j', X, X)) + eps
    X /= norms[np.newaxis, :]
    return X


# --------------------------------------------------------------------------- batching
def gen_even_batches(N, n_batches):
    """lyssa/utils/__init__.py:166-180 -- n_batches-1 batches of floor(N/n_batches), last takes the rest."""
    size = int(np.floor(N / float(n_batches)))
    out, base = [], 0
    for _ in range(n_batches - 1):
        out.append(range(base, base + size))
        base += size
    out.append(range(base, N))
    return out


def gen_batches(N, batch_size=None):
    """lyssa/utils/__init__.py:183-201 -- fixed-size batches + remainder; None => one batch."""
    if batch_size is None:
        return [range(0, N)]
    out, base = [], 0
    for _ in range(int(np.floor(N / float(batch_size)))):
        out.append(range(base, base + batch_size))
        base += batch_size
    if N > base:
        out.append(range(base, N))
    return out


# --------------------------------------------------------------------------- Batch-OMP
def batch_omp_signal(a0, G, k, want_gap=False):
    """One signal of lyssa/sparse_coding.py:310-365.

    Returns (support list in selection order, coefficient array, min relative top1/top2 gap).
    Greedy loop: argmax|a| (first max wins) :322; break on re-selection :323-325;
    j==0: z=a0[kk] :360-363; j==1: closed-form 2x2 factor :330-338; j>=2: w=L^-1 g,
    vs=1-w'w, break if vs<eps :340-349; two triangular solves :353-354; a=a0-G[:,Dx]z :359.
    The Gram diagonal is *assumed* to be 1 (hard-coded), exactly as the reference does.
    """
    K = a0.shape[0]
    L = np.zeros((k, k))
    Dx = []
    z = np.zeros(0)
    a = a0
    min_gap = np.inf
    for j in range(k):
        absa = np.abs(a)
        kk = int(np.argmax(absa))
        if want_gap and K > 1:
            top = absa[kk]
            tmp = absa.copy()
            tmp[kk] = -1.0
            second = tmp.max()
            gap = (top - second) / top if top > 0 else 0.0
            if kk not in Dx:
                min_gap = min(min_gap, gap)
        if kk in Dx:
            break
        if j == 0:
            Dx.append(kk)
            z = a0[Dx].copy()
        else:
            g = G[Dx, kk]
            if j == 1:
                w = float(g[0])
                vs = 1.0 - w * w
                if vs < EPS64:
                    break
                L[0, 0] = 1.0
                L[1, 0] = w
                L[1, 1] = np.sqrt(vs)
            else:
                w = solve_triangular(L[:j, :j], g, lower=True, check_finite=False)
                vs = 1.0 - float(np.dot(w, w))
                if vs < EPS64:
                    break
                L[j, :j] = w
                L[j, j] = np.sqrt(vs)
            Dx.append(kk)
            Ltc = solve_triangular(L[:j + 1, :j + 1], a0[Dx], lower=True)
            z = solve_triangular(L[:j + 1, :j + 1], Ltc, trans=1, lower=True)
        a = a0 - np.dot(G[:, Dx], z)
    return Dx, z, (min_gap if np.isfinite(min_gap) else 0.0)


def batch_omp(X, Alpha, D, Gram, n_nonzero_coefs=None, tol=None):
    """lyssa/sparse_coding.py:302-367 -- dense Z (n_atoms, n_samples). X, D only give shapes; tol unused."""
    n_samples = X.shape[1]
    n_atoms = D.shape[1]
    Z = np.zeros((n_atoms, n_samples))
    for i in range(n_samples):
        Dx, z, _ = batch_omp_signal(Alpha[:, i], Gram, n_nonzero_coefs)
        Z[Dx, i] = z
    return Z


def bomp_encode(X, D, k):
    """sparse_encoder.__call__, 'bomp' branch: lyssa/sparse_coding.py:629-635 + :718-720 (n_jobs=1)."""
    Gram = fast_dot(D.T, D)
    Alpha = fast_dot(D.T, X)
    return batch_omp(X, Alpha, D, Gram, n_nonzero_coefs=k)


def bomp_encode_sparse(X, D, k):
    """Same computation, returned as the sparse triplet the HIP engine produces.

    idx (N,k) int32 selection order, -1 padded; coef (N,k) float64, 0 padded; nnz (N,) int32 = len(Dx);
    gap (N,) float64 = min relative top-1/top-2 gap along the greedy path (tie classifier).
    """
    Gram = fast_dot(D.T, D)
    Alpha = fast_dot(D.T, X)
    N = X.shape[1]
    idx = -np.ones((N, k), dtype=np.int32)
    coef = np.zeros((N, k))
    nnz = np.zeros(N, dtype=np.int32)
    gap = np.zeros(N)
    for i in range(N):
        Dx, z, g = batch_omp_signal(Alpha[:, i], Gram, k, want_gap=True)
        m = len(Dx)
        idx[i, :m] = Dx
        coef[i, :m] = z
        nnz[i] = m
        gap[i] = g
    return idx, coef, nnz, gap


def densify(idx, coef, nnz, K):
    """Sparse triplet -> dense float64 Z (K, N), the reference's return type (sparse_coding.py:365)."""
    N = idx.shape[0]
    Z = np.zeros((K, N))
    for i in range(N):
        m = int(nnz[i])
        Z[idx[i, :m], i] = coef[i, :m]
    return Z


# --------------------------------------------------------------------------- 'omp' and 'thresh' 
def omp_signal(x, D, Gram, alpha, n_nonzero_coefs, want_gap=False, tol=None):
    """lyssa/sparse_coding.py:19-57 (`_omp`, fixed sparsity): residual-domain OMP.  argmax|alpha| :39, stop on
    re-selection :40-41, z[Dx] = inv(G[Dx,Dx]) (D'x)[Dx] :44-52 (TRUE Gram diagonal, unlike batch_omp),
    r = x - D[:,Dx] z :53, alpha = D'r :54; loop while i < k and ||r|| > 1e-10 :27-31.
    ``want_gap`` also returns the minimum relative top-1/top-2 gap of |alpha| along the greedy path (tie classifier of
    the parity tests, same definition as batch_omp_signal).  ``n_nonzero_coefs=None`` with ``tol``: the error-constrained
    form, loop while ||r|| >= tol :30-31."""
    K = D.shape[1]
    Dx = []
    z = np.zeros(K)
    r = np.copy(x)
    i = 0
    a0 = np.dot(D.T, x)
    min_gap = np.inf
    def cont():
        if n_nonzero_coefs is not None:
            return i < n_nonzero_coefs and norm(r) > 1e-10
        return norm(r) >= tol

    while cont():
        kk = int(np.argmax(np.abs(alpha)))
        if kk in Dx:
            break
        if want_gap and K > 1:
            two = np.partition(np.abs(alpha), K - 2)[K - 2:]
            min_gap = min(min_gap, (two[1] - two[0]) / two[1] if two[1] > 0 else 0.0)
        Dx.append(kk)
        Gs = np.atleast_2d(Gram[Dx, :][:, Dx])
        try:
            Gi = np.linalg.inv(Gs)
        except np.linalg.LinAlgError:
            break
        z[Dx] = np.dot(Gi, a0[Dx])
        r = x - np.dot(D[:, Dx], z[Dx])
        alpha = np.dot(D.T, r)
        i += 1
    if want_gap:
        return z, (min_gap if np.isfinite(min_gap) else 0.0)
    return z


def omp_encode(X, D, k, want_gap=False, tol=None):
    """sparse_encoder 'omp' branch: lyssa/sparse_coding.py:620-627 + `omp` :60-66.  ``want_gap`` -> (Z, gap [N]);
    ``k=None`` with ``tol``: error-constrained."""
    Gram = fast_dot(D.T, D)
    Alpha = fast_dot(D.T, X)
    Z = np.zeros((D.shape[1], X.shape[1]))
    gap = np.zeros(X.shape[1])
    for i in range(X.shape[1]):
        if want_gap:
            Z[:, i], gap[i] = omp_signal(X[:, i], D, Gram, Alpha[:, i], k, want_gap=True, tol=tol)
        else:
            Z[:, i] = omp_signal(X[:, i], D, Gram, Alpha[:, i], k, tol=tol)
    return (Z, gap) if want_gap else Z


def thresh_gap(Alpha, n_nonzero_coefs):
    """Tie classifier of 'thresh': relative gap between the k-th and the (k+1)-th largest SIGNED correlation of every
    column (the only place where rounding can change the selected set of sparse_coding.py:416-425)."""
    K, N = Alpha.shape
    if n_nonzero_coefs >= K:
        return np.full(N, np.inf)
    srt = np.sort(Alpha, axis=0)[::-1]
    a, b = srt[n_nonzero_coefs - 1], srt[n_nonzero_coefs]
    return (a - b) / np.maximum(np.abs(Alpha).max(axis=0), 1e-300)


def thresholding(Alpha, nonzero_percentage=None, n_nonzero_coefs=None):
    """lyssa/sparse_coding.py:416-425: keep the k largest SIGNED correlations of every column."""
    K, N = Alpha.shape
    Z = np.zeros((K, N))
    if nonzero_percentage is not None:
        n_nonzero_coefs = int(np.floor(nonzero_percentage * K))
    for i in range(N):
        idx = Alpha[:, i].argsort()[::-1][:n_nonzero_coefs]
        Z[idx, i] = Alpha[idx, i]
    return Z


def thresh_encode(X, D, n_nonzero_coefs=None, nonzero_percentage=None):
    """sparse_encoder 'thresh' branch: lyssa/sparse_coding.py:637-642."""
    return thresholding(fast_dot(D.T, X), nonzero_percentage=nonzero_percentage, n_nonzero_coefs=n_nonzero_coefs)


# --------------------------------------------------------------------------- dictionary helpers
def lasso_signal(a0, G, lam, tol=1e-13, max_steps=100000):
    """argmin_a 0.5||x - D a||^2 + lam ||a||_1 given a0 = D'x and G = D'D: greedy coordinate descent, float64.

    The reference delegates to spams.lasso(mode=2, lambda2=0) (sparse_coding.py:487-509); SPAMS is absent, so this
    restates the PROBLEM, not SPAMS' LARS code ("lasso parity unpinned") -- it is pinned against sklearn's
    independent solvers and the KKT conditions in tests/test_oracle_golden.py."""
    K = a0.shape[0]
    a = np.zeros(K)
    c = a0.astype(np.float64).copy()
    gd = np.diag(G).astype(np.float64)
    ginv = np.where(gd > 0, 1.0 / np.where(gd > 0, gd, 1.0), 0.0)
    scale = np.max(np.abs(a0)) if K else 0.0
    for _ in range(max_steps):
        v = c + gd * a
        s = np.sign(v) * np.maximum(np.abs(v) - lam, 0.0) * ginv
        d = s - a
        j = int(np.argmax(np.abs(d)))
        if not (abs(d[j]) > tol * scale):
            break
        a[j] += d[j]
        c -= d[j] * G[:, j]
    return a


def lasso_encode(X, D, lam, tol=1e-13):
    """Dense (K, N) lasso codes, one column per signal (sparse_coding.py:697-698)."""
    G = fast_dot(D.T, D)
    A0 = fast_dot(D.T, X)
    Z = np.zeros((D.shape[1], X.shape[1]))
    for i in range(X.shape[1]):
        Z[:, i] = lasso_signal(A0[:, i], G, lam, tol=tol)
    return Z


def lasso_kkt_violation(X, D, Z, lam):
    """max over signals/atoms of the KKT residual of min 0.5||x-Da||^2 + lam||a||_1, relative to max|D'x|:
    |d_j'(x - D a)| <= lam where a_j = 0, and d_j'(x - D a) = lam sign(a_j) where a_j != 0."""
    C = fast_dot(D.T, X - fast_dot(D, Z))
    scale = np.maximum(np.max(np.abs(fast_dot(D.T, X)), axis=0), 1e-300)
    zero = Z == 0
    v = np.where(zero, np.maximum(np.abs(C) - lam, 0.0), np.abs(C - lam * np.sign(Z)))
    return float(np.max(v / scale[None, :]))


def approx_error(D, Z, X):
    """lyssa/dict_learning/utils.py:14-19 -- ||X - DZ||_F^2."""
    return frobenius_squared(X - fast_dot(D, Z))


def average_mutual_coherence(D):
    """lyssa/dict_learning/utils.py:7-11 -- mean off-diagonal |D'D|."""
    K = D.shape[1]
    G = np.abs(np.dot(D.T, D))
    np.fill_diagonal(G, 0)
    return float(np.sum(G) / float(K * (K - 1)))


def force_mi(D, X, Z, unused_data, eta, max_tries=100):
    """lyssa/dict_learning/utils.py:86-139, statement by statement (same global-RNG draws).  The two paths on which the
    reference itself fails are made explicit: `min_idx is None` at :134 (no candidate lowered the coherence; the
    reference would index X[:, None]) leaves the atom alone, and an exhausted candidate list (:119-120, bare `return D`)
    returns (D, unused_data)."""
    n_atoms = D.shape[1]
    G = np.abs(np.dot(D.T, D))                                   # :88 (computed once, never refreshed)
    np.fill_diagonal(G, 0)                                       # :89
    for atom_idx1 in range(n_atoms):                             # :91
        atom_idx2 = np.argmax(G[atom_idx1, :])                   # :93
        mcoh = G[atom_idx1, atom_idx2]                           # :95
        if mcoh < eta:                                           # :96
            continue
        if norm(Z[atom_idx1, :]) > norm(Z[atom_idx2, :]):        # :101
            c_atom = atom_idx1
        else:
            c_atom = atom_idx2
        cnt = 0
        available_data = unused_data[:]                          # :108
        min_idx = None
        min_coh = mcoh
        while mcoh > eta:                                        # :112
            if cnt > max_tries:                                  # :114
                break
            if len(available_data) == 0:                         # :117
                return D, unused_data
            idx = np.random.choice(available_data, size=1)[0]    # :119,122
            new_atom = normalize(X[:, idx])                      # :123-124
            available_data.remove(idx)                           # :125
            g = np.abs(np.dot(D.T, new_atom))                    # :126
            mcoh = np.max(g)                                     # :127
            if mcoh < min_coh:                                   # :128
                min_coh = mcoh
                min_idx = idx
            cnt += 1
        if min_idx is None:
            continue
        D[:, c_atom] = normalize(X[:, min_idx])                  # :134-135
        unused_data.remove(min_idx)                              # :136
    return D, unused_data


def init_dictionary(X, n_atoms, method='data', return_unused_data=False, normalize=True):
    """lyssa/dict_learning/utils.py:49-70 ('data' method only).

    Candidates = columns with energy > 1e-6; np.random.choice on the GLOBAL RNG, no replacement;
    D = X[:, chosen] (copy) optionally norm_cols'ed; unused_data = remaining candidates (list).
    """
    if method != 'data':
        raise ValueError("oracle restates only method='data'")
    n_samples = X.shape[1]
    idxs = [i for i in range(n_samples) if np.sum(X[:, i] ** 2) > 1e-6]
    if len(idxs) < n_atoms:
        raise ValueError("not enough datapoints to initialize the dictionary")
    subset = np.random.choice(len(idxs), size=n_atoms, replace=False)
    subset_idxs = np.array(idxs).astype(int)[subset]
    D = X[:, subset_idxs]
    if normalize:
        D = norm_cols(D)
    if return_unused_data:
        s = set(subset_idxs)
        return D, [x for x in idxs if x not in s]
    return D


# --------------------------------------------------------------------------- approximate K-SVD
def approx_ksvd(Y, D, X, n_cycles=1):
    """lyssa/dict_learning/ksvd.py:98-126.  Mutates D and X in place; returns (D, X, unused_atoms).

    R = Y - DX :103; atoms visited in order 0..K-1 (Gauss-Seidel through R) :105-106;
    omega = X[k,:]!=0, skip+record when empty :111-115; Rk = R[:,omega] + d_k x_k :116;
    d_k = normalize(Rk x_k) :118-119; x_k = Rk' d_k :121; R[:,omega] = Rk - d_k x_k :123.
    """
    n_atoms = D.shape[1]
    unused = []
    R = Y - fast_dot(D, X)
    for _ in range(n_cycles):
        for k in range(n_atoms):
            omega = X[k, :] != 0
            if not np.any(omega):
                unused.append(k)
                continue
            xk = X[k, omega]
            Rk = R[:, omega] + np.outer(D[:, k], xk)
            D[:, k] = normalize(np.dot(Rk, xk))
            X[k, omega] = np.dot(Rk.T, D[:, k])
            R[:, omega] = Rk - np.outer(D[:, k], X[k, omega])
    return D, X, unused


def ksvd_exact(Y, D, X, n_cycles=1):
    """lyssa/dict_learning/ksvd.py:19-43 (`ksvd`).  Mutates D and X in place; returns (D, X, unused_atoms).

    The reference takes U,S,V = randomized_svd(Rk, n_components=1, n_iter=10, flip_sign=False) :35 -- a randomized
    range finder whose sign is arbitrary and whose accuracy depends on the spectrum.  The restatement uses the EXACT
    leading singular triplet (numpy SVD), sign chosen so that u . d_old >= 0; parity with the reference is therefore
    up to the sign of (d_k, x_k) and the randomized solver's accuracy (tests/test_oracle_golden.py, F10).
    """
    n_atoms = D.shape[1]
    unused = []
    R = Y - fast_dot(D, X)
    for _ in range(n_cycles):
        for k in range(n_atoms):
            omega = X[k, :] != 0
            if not np.any(omega):
                unused.append(k)
                continue
            Rk = R[:, omega] + np.outer(D[:, k], X[k, omega])
            U, S, Vt = np.linalg.svd(Rk, full_matrices=False)
            u, sv = U[:, 0], S[0] * Vt[0, :]
            if np.dot(u, D[:, k]) < 0:
                u, sv = -u, -sv
            D[:, k] = u
            X[k, omega] = sv
            R[:, omega] = Rk - np.outer(u, sv)
    return D, X, unused


def nn_ksvd(Y, D, X, n_cycles=1, signs=None):
    """lyssa/dict_learning/ksvd.py:46-95 (`nn_ksvd`).  Mutates D and X in place; returns (D, X, unused_atoms).

    One pass over the atoms (no outer cycle loop, :53); per atom the rank-1 SVD of Rk :66, d = max(u, 0), x = max(S v, 0)
    :73-77, `continue` when d'd or x'x <= eps :79-82 (atom, codes and residual untouched), n_cycles alternating projections
    d = max(Rk x / x'x, 0), x = max(d'Rk / d'd, 0) :84-88, then d /= ||d||, x *= ||d|| :90-92 and the residual update :95.
    The reference's randomized_svd(n_iter=50, flip_sign=False) returns (u, v) with an ARBITRARY common sign, and the clip
    makes the result depend on it.  The restatement uses the exact SVD with u . d_old >= 0; `signs` (one +-1 per USED atom, in
    visiting order -- what the reference's solver happened to return, recorded by oracle/make_golden.py) reproduces the
    reference's own run instead.
    """
    n_atoms = D.shape[1]
    unused = []
    R = Y - fast_dot(D, X)
    used = 0
    for k in range(n_atoms):
        omega = X[k, :] != 0
        if not np.any(omega):
            unused.append(k)
            continue
        Rk = R[:, omega] + np.outer(D[:, k], X[k, omega])
        U, S, Vt = np.linalg.svd(Rk, full_matrices=False)
        d, x = U[:, 0].copy(), S[0] * Vt[0, :]
        sgn = 1.0 if np.dot(d, D[:, k]) >= 0 else -1.0
        if signs is not None:
            sgn *= float(signs[used])
        used += 1
        d, x = sgn * d, sgn * x
        d[d < 0] = 0
        x[x < 0] = 0
        if np.dot(d, d) <= EPS64 or np.dot(x, x) <= EPS64:
            continue
        for _ in range(n_cycles):
            d = np.dot(Rk, x) / np.dot(x, x)
            d[d < 0] = 0
            x = np.dot(d.T, Rk) / np.dot(d, d)
            x[x < 0] = 0
        nrm = norm(d)
        d = d / nrm
        x = x * nrm
        D[:, k] = d
        X[k, omega] = x
        R[:, omega] = Rk - np.outer(D[:, k], X[k, omega])
    return D, X, unused


def ksvd_dict_learn(X, n_atoms, init_dict='data', encode=None, max_iter=20, n_cycles=1, verbose=True,
                    trace=None):
    """lyssa/dict_learning/ksvd.py:129-231, approx=True, eta=None path.

    ``encode(X, D) -> dense Z`` plays the role of ``sparse_coder``.  Reproduces the host control
    flow including the patience quirk (:222-229): error_prev is only refreshed when verbose, and
    then *before* the test, so patience increments on every iteration it>0 and the loop stops
    after 11 iterations whatever max_iter is.  Unused atoms are replaced from ``unused_data``
    using the GLOBAL numpy RNG (:199-207).  ``trace`` (list) receives per-iteration dicts.
    """
    unused_data = []
    if isinstance(init_dict, str) and init_dict == 'data':
        D, unused_data = init_dictionary(X, n_atoms, method='data', return_unused_data=True)
    else:
        D = np.copy(init_dict)
    max_patience = 10
    error_curr = 0
    error_prev = 0
    it = 0
    patience = 0
    Z = np.zeros((n_atoms, X.shape[1]))
    while it < max_iter and patience < max_patience:
        Z = encode(X, D)
        D, _, unused_atoms = approx_ksvd(X, D, Z, n_cycles=n_cycles)
        for j in range(len(unused_atoms)):
            if len(unused_data) == 0:
                break
            idx = np.random.choice(unused_data, size=1)[0]
            D[:, unused_atoms[j]] = X[:, idx]
            D[:, unused_atoms[j]] = normalize(D[:, unused_atoms[j]])
            unused_data.remove(idx)
        error_curr = approx_error(D, Z, X)
        if trace is not None:
            trace.append(dict(it=it, D=D.copy(), error=error_curr, unused_atoms=list(unused_atoms)))
        if verbose:
            error_prev = error_curr
        if (it > 0) and (error_curr > 0.9 * error_prev or error_curr > error_prev):
            patience += 1
        it += 1
    return D, Z


# --------------------------------------------------------------------------- online dictionary learning
def odl_batch_update(D, A, B, X_batch, Z_batch, beta_i, non_neg=False):
    """One mini-batch of lyssa/dict_learning/online_dict_learn.py:84-98.

    A = beta*A + ZZ' :84; B = beta*B + XZ' :85; DA = D A computed ONCE :91;
    d_k += (B_k - DA_k)/(A_kk+eps) for every k (Jacobi-style) :93-94; clip :96-97; norm_cols :98.
    Returns new (D, A, B); D is updated in place like the reference.
    """
    A = beta_i * A + fast_dot(Z_batch, Z_batch.T)
    B = beta_i * B + fast_dot(X_batch, Z_batch.T)
    DA = fast_dot(D, A)
    for k in range(D.shape[1]):
        D[:, k] = (1 / (A[k, k] + EPS64)) * (B[:, k] - DA[:, k]) + D[:, k]
    if non_neg:
        D[D < 0] = 0
    D = norm_cols(D)
    return D, A, B


def online_dict_learn(X, n_atoms, encode=None, batch_size=None, A=None, B=None, D_init=None,
                      beta=None, n_epochs=1, verbose=False, non_neg=False, trace=None):
    """lyssa/dict_learning/online_dict_learn.py:18-124 (host control flow + batch update).

    beta=None => linspace(0,1,n_iter) restarted every epoch (:65-67,79) -- beta[0]=0 wipes A,B.
    Epoch-end error pass and the same patience quirk as K-SVD (:101-118).
    """
    n_features, n_samples = X.shape
    if D_init is None:
        D, _ = init_dictionary(X, n_atoms, method='data', return_unused_data=True)
    else:
        D = D_init
    batch_idx = gen_batches(n_samples, batch_size=batch_size)
    n_iter = len(batch_idx)
    if A is None and B is None:
        A = np.zeros((n_atoms, n_atoms))
        B = np.zeros((n_features, n_atoms))
    if beta is None:
        beta = np.linspace(0, 1, num=n_iter)
    else:
        beta = np.zeros(n_iter) + beta
    max_patience = 10
    error_curr = 0
    error_prev = 0
    patience = 0
    for e in range(n_epochs):
        for i, batch in zip(range(n_iter), batch_idx):
            Xb = X[:, batch]
            Zb = encode(Xb, D)
            D, A, B = odl_batch_update(D, A, B, Xb, Zb, beta[i], non_neg=non_neg)
            if trace is not None:
                trace.append(dict(epoch=e, batch=i, D=D.copy(), A=A.copy(), B=B.copy()))
        if e < n_epochs - 1:
            if patience >= max_patience:
                return D, A, B
            error_curr = 0
            for i, batch in zip(range(n_iter), batch_idx):
                Xb = X[:, batch]
                Zb = encode(Xb, D)
                error_curr += approx_error(D, Zb, Xb)
            if verbose:
                error_prev = error_curr
            if (e > 0) and (error_curr > 0.9 * error_prev or error_curr > error_prev):
                patience += 1
    return D, A, B


# --------------------------------------------------------------------------- projected gradient descent
def pgd_batch_update(D, X_batch, Z_batch, eta, mu=None, non_neg=False):
    """One mini-batch of lyssa/dict_learning/gradient_descent.py:84-98 (grad :85, incoherence term ADDED :92,
    clip :94-95, norm_cols :97)."""
    grad_approx = np.dot(np.dot(D, Z_batch) - X_batch, Z_batch.T)
    if mu is not None and mu > 0:
        grad_incoh = 2 * mu * np.dot(D, np.dot(D.T, D) - np.eye(D.shape[1]))
    else:
        grad_incoh = 0
    D = D - (eta * grad_approx) + grad_incoh
    if non_neg:
        D[D < 0] = 0
    return norm_cols(D)


# --------------------------------------------------------------------------- patches / preprocessing
def grid_patches(img, patch_size, step_size):
    """lyssa/utils/img.py:420-477 (no random subset): (patch_size^2 * C, n_patches), patches in row-major grid order,
    features = C-order flatten of (patch_size, patch_size, C)."""
    img = np.asarray(img)
    if img.ndim == 2:
        img = img[:, :, None]
    H, W, C = img.shape
    n_ph = (H - patch_size) // step_size + 1
    n_pw = (W - patch_size) // step_size + 1
    out = np.zeros((patch_size * patch_size * C, n_ph * n_pw), dtype=img.dtype)
    for i in range(n_ph):
        for j in range(n_pw):
            out[:, i * n_pw + j] = img[i * step_size:i * step_size + patch_size,
                                       j * step_size:j * step_size + patch_size, :].reshape(-1)
    return out


def preproc(name, X):
    """lyssa/feature_extract/preproc.py:46-80 (per-datapoint and dataset-level operations)."""
    X = np.array(X, dtype=np.float64)
    if name == 'scaling':
        return X / 255.
    if name == 'local_centering':
        return X - X.mean(axis=0)[np.newaxis, :]
    if name == 'contrast_normalization':
        return norm_cols(X - X.mean(axis=0)[np.newaxis, :])
    if name == 'normalization':
        return norm_cols(X)
    if name == 'global_centering':           # :55-57, per FEATURE over all datapoints
        return X - X.mean(axis=1)[:, np.newaxis]
    if name == 'global_standarization':      # :58-62
        X = X - X.mean(axis=1)[:, np.newaxis]
        return X / X.std(axis=1)[:, np.newaxis]
    if name == 'whitening':                  # :77-78 -> zca_transform(X.T).T, :18-31 (bias = 0.1)
        Xr = X.T - X.T.mean(axis=0)
        eigs, eigv = np.linalg.eigh(np.dot(Xr.T, Xr) / Xr.shape[0] + 0.1 * np.identity(Xr.shape[1]))
        return np.dot(Xr, np.dot(eigv * np.sqrt(1.0 / eigs), eigv.T)).T
    raise ValueError(name)


# --------------------------------------------------------------------------- ScSPM pooling
def spm_pool(Z, pos, patch_size, imsize, levels=(1, 2, 4), l2=False):
    """Pooling loop of lyssa/feature_extract/spatial_pyramid.py:57-97 with sc_max_pooling (pooling.py:4-7) and the
    optional l2_normalizer (preproc.py:8-16): dense codes Z (K, n_patches) of one image -> flattened pyramid."""
    py, px = pos[:, 0], pos[:, 1]
    cy = py + float(patch_size) / 2 - 0.5
    cx = px + float(patch_size) / 2 - 0.5
    K = Z.shape[0]
    n_cells = int(np.sum(np.array(levels) ** 2))
    pooled = np.zeros((n_cells, K))
    cnt = 0
    for lev in levels:
        wunit = float(imsize[1]) / lev
        hunit = float(imsize[0]) / lev
        binidx = np.floor(cy / hunit) * lev + np.floor(cx / wunit)
        for j in range(lev * lev):
            pidx = np.nonzero(binidx == j)[0]
            if len(pidx) > 0:
                pooled[cnt, :] = np.max(np.abs(Z[:, pidx]), axis=1)
                if l2:
                    pooled[cnt, :] = normalize(pooled[cnt, :])
            cnt += 1
    return pooled.flatten()
