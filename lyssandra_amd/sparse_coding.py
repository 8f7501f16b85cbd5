"""Drop-in for ``lyssa.sparse_coding.sparse_encoder`` (lyssa/sparse_coding.py:512-726) on MI355X.

Same constructor, attributes and call contract as the reference class:

    se = sparse_encoder(algorithm='bomp', params={'n_nonzero_coefs': 10}, n_jobs=1)
    Z = se.encode(X, D)          # X (n_features, n_samples), D (n_features, n_atoms)
                                 # -> NEW dense float64 (n_atoms, n_samples), like the reference

``'bomp'`` runs entirely in liblyssa_hip.so (fp32 MFMA GEMMs for G = D'D and alpha0 = D'X, wave-per-signal
greedy/Cholesky kernel); ``'omp'`` (fixed ``n_nonzero_coefs``) and ``'thresh'`` ride on the same engine (SURVEY 8f
rank 1), and so does ``'lasso'`` (``params['lambda']``; Gram-based coordinate descent instead of SPAMS' LARS, same
minimiser -- SURVEY 8f rank 4).  There is NO CPU fallback: without the library or without a GPU the call raises.
Unknown algorithms raise ``ValueError("Sparse optimizer not found.")`` exactly like sparse_coding.py:705-706;
the reference's other algorithms are outside the accelerated path and raise NotImplementedError.

Beyond the reference API, ``encode_sparse`` returns the device-resident sparse triplet (idx, coef, nnz) --
the dense float64 (K, N) return type costs 8 KB per signal at K=1024 and is only materialised on request.
"""
import os
import tempfile

import numpy as np

from . import engine

_REFERENCE_ALGORITHMS = ('omp', 'bomp', 'thresh', 'nnomp', 'group_omp', 'sparse_group_omp', 'somp', 'iht',
                         'lasso', 'llc')


class sparse_encoder(object):
    """MI355X implementation of the reference's sparse_encoder (only ``algorithm='bomp'`` is accelerated)."""

    def __init__(self, algorithm='omp', params=None, n_jobs=1, verbose=True, mmap=False, name='sparse_coder', n_gpus=1):
        # lyssa/sparse_coding.py:587-598
        self.name = name
        self.algorithm = algorithm
        self.params = params
        if self.params is None:
            self.params = {}
        if n_jobs == -1:
            from .utils import cpu_count
            n_jobs = cpu_count
        self.n_jobs = n_jobs      # kept for API compatibility; parallelism comes from the GPU(s)
        self.verbose = verbose
        self.mmap = mmap
        self.device = None        # None => current HIP device
        # n_gpus > 1 (or -1 = every visible device): ONE process shards the columns of X over the devices the way the
        # reference's run_parallel(n_jobs=N) shards them over worker processes (lyssa/sparse_coding.py:713-724,
        # lyssa/utils/__init__.py:92-129) -- through the library-owned multi-device context (lys_ctx_create_multi)
        self.n_gpus = n_gpus
        self._ctx_devices = None  # test hook: explicit device list for the context path (also with one device)
        self._dd = None           # cached DeviceDictionary (re-packed on every call: D may have changed)

    # -- reference API ---------------------------------------------------------------------------------
    def encode(self, X, D):
        return self.__call__(X, D)

    def __call__(self, X, D):
        """lyssa/sparse_coding.py:603-726 -> dense float64 Z (n_atoms, n_samples)."""
        n_samples = X.shape[1]
        n_atoms = D.shape[1]
        if self.params.get('lambda') is not None:
            assert self.params.get('lambda') <= n_atoms
        self._check_algorithm()
        devs = self._multi_devices()
        if devs is not None:
            return self._encode_multi(X, D, devs)
        idx, coef, nnz = self.encode_sparse(X, D)
        out = _empty_mmap((n_atoms, n_samples)) if self.mmap else None
        return engine.densify(idx, coef, nnz, n_atoms, out=out)

    def _multi_devices(self):
        if self._ctx_devices is not None:
            return list(self._ctx_devices)
        if self.n_gpus in (None, 0, 1) or self.algorithm != 'bomp':
            return None
        import torch
        n_dev = torch.cuda.device_count()
        want = n_dev if self.n_gpus == -1 else int(self.n_gpus)
        if want > n_dev:
            raise ValueError("n_gpus=%d but only %d HIP device(s) are visible" % (want, n_dev))
        return list(range(want)) if want > 1 else None

    def _encode_multi(self, X, D, devices):
        """'bomp' over several devices in this one process: host arrays in, dense float64 (K, N) out."""
        import ctypes
        from . import _lib
        lib = _lib.load()
        engine.require_gpu()
        n, N = X.shape
        K = D.shape[1]
        k = self._k(K)
        Xh = np.ascontiguousarray(np.asarray(X).T, dtype=np.float32)       # signal-major [N][n]
        Dh = np.ascontiguousarray(np.asarray(D).T, dtype=np.float32)       # atom-major [K][n]
        idx = np.empty((N, k), dtype=np.int32)
        coef = np.empty((N, k), dtype=np.float32)
        nnz = np.empty((N,), dtype=np.int32)
        P = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
        ids = (ctypes.c_int * len(devices))(*devices)
        ctx = ctypes.c_void_p()
        _lib.check(lib.lys_ctx_create_multi(len(devices), ids, ctypes.byref(ctx)), "lys_ctx_create_multi")
        try:
            _lib.check(lib.lys_ctx_set_dictionary(ctx, P(Dh), n, K), "lys_ctx_set_dictionary")
            _lib.check(lib.lys_ctx_bomp_encode(ctx, P(Xh), N, k, P(idx), P(coef), P(nnz)), "lys_ctx_bomp_encode")
        finally:
            lib.lys_ctx_destroy(ctx)
        Z = _empty_mmap((K, N)) if self.mmap else np.zeros((K, N))
        if self.mmap:
            Z[:] = 0.0
        cols = np.arange(N)
        for j in range(k):
            m = nnz > j
            Z[idx[m, j], cols[m]] = coef[m, j]
        return Z

    # -- extended API ----------------------------------------------------------------------------------
    def encode_sparse(self, X, D):
        """Same inputs as ``encode``; returns the device-resident triplet (idx [N,k], coef [N,k], nnz [N])."""
        self._check_algorithm()
        Xs = engine.signals_to_device(X, self.device)
        dd = self._dictionary(D)
        return self.encode_device(Xs, dd)

    def encode_device(self, Xs, dd, out=None):
        """Device-resident form: ``Xs`` signal-major fp32 cuda tensor [N, n], ``dd`` an engine.DeviceDictionary;
        ``out`` = a previously returned triplet to overwrite (stable buffers for iterative learners)."""
        self._check_algorithm()
        if self.algorithm == 'lasso':
            # sparse_coding.py:697-698: lasso(params['lambda'], n_jobs)(X, D); extra knobs: kcap / max_steps / tol
            lam = self.params.get('lambda')
            if lam is None:
                raise ValueError("params['lambda'] is required for algorithm='lasso'")
            # params['solver']: 'lars' (LARS-lasso homotopy like SPAMS + coordinate-descent polish, the default) or 'cd'
            idx, coef, nnz, steps = engine.lasso_encode(Xs, dd, lam, kcap=self.params.get('kcap'),
                                                        max_steps=self.params.get('max_steps'),
                                                        tol=self.params.get('tol', 1e-6), out=out, return_steps=True,
                                                        solver=self.params.get('solver', 'lars'))
            if steps.numel():
                if int(steps.min().item()) < 0:
                    raise RuntimeError("lasso: more than kcap=%d non-zero coefficients for some signal; raise "
                                       "params['kcap'] or lambda" % idx.shape[1])
                budget = self.params.get('max_steps')
                budget = int(budget) if budget is not None else 50 * int(idx.shape[1])
                n_bad = int((steps >= budget).sum().item())
                if n_bad:
                    import warnings
                    warnings.warn("lasso: %d of %d signals used the whole step budget (%d) without reaching tol; "
                                  "raise params['max_steps'] or lambda" % (n_bad, steps.numel(), budget))
            return idx, coef, nnz
        if self.algorithm == 'omp' and self.params.get('n_nonzero_coefs') is None and self.params.get('tol') is not None:
            # error-constrained OMP (sparse_coding.py:27-31): atoms until ||r|| < tol; at most params['kcap'] (<= 64) of them
            kcap = self.params.get('kcap')
            idx, coef, nnz = engine.omp_tol_encode(Xs, dd, self.params.get('tol'), kcap=kcap, out=out)
            cap = int(idx.shape[1])
            if cap < min(dd.n, dd.K) and nnz.numel() and int(nnz.max().item()) >= cap:
                import warnings
                warnings.warn("omp(tol): some signals used all %d coefficient slots; the reference would keep adding "
                              "atoms (params['kcap'] <= 64 sets the limit)" % cap)
            return idx, coef, nnz
        return engine.bomp_encode(Xs, dd, self._k(dd.K), out=out, algorithm=self.algorithm)

    # -- helpers ---------------------------------------------------------------------------------------
    def _k(self, n_atoms=None):
        k = self.params.get('n_nonzero_coefs')
        if self.algorithm == 'thresh' and self.params.get('nonzero_percentage') is not None:
            # thresholding(): nonzero_percentage overrides n_nonzero_coefs (sparse_coding.py:419-420)
            k = int(np.floor(self.params.get('nonzero_percentage') * n_atoms))
        if k is None:
            # the reference dies with a TypeError in np.zeros((None, None)) (sparse_coding.py:317)
            raise ValueError("params['n_nonzero_coefs'] is required for algorithm=%r" % (self.algorithm,))
        return int(k)

    def _check_algorithm(self):
        if self.algorithm in ('bomp', 'omp', 'thresh', 'lasso'):
            return
        if self.algorithm in _REFERENCE_ALGORITHMS:
            raise NotImplementedError(
                "algorithm=%r is outside the MI355X-accelerated path ('bomp', 'omp', 'thresh', 'lasso' are implemented; "
                "there is deliberately no CPU fallback in this package)" % (self.algorithm,))
        raise ValueError("Sparse optimizer not found.")  # sparse_coding.py:705-706

    def _dictionary(self, D):
        """Packed device copy of D + its Gram matrix, re-used while the CONTENT of D is unchanged (a 64-bit hash of the
        host array; a learner that edits D in place between calls therefore still gets a fresh upload)."""
        D = np.asarray(D) if not _is_tensor(D) else D
        n, K = int(D.shape[0]), int(D.shape[1])
        dd = self._dd
        if dd is None or dd.n != n or dd.K != K:
            dd = engine.DeviceDictionary(n, K, self.device)
            self._dd = dd
            self._dd_tag = None
        tag = None
        if not _is_tensor(D):
            try:
                import xxhash
                tag = (D.dtype.str, xxhash.xxh3_64_intdigest(np.ascontiguousarray(D)))
            except Exception:  # pragma: no cover
                tag = None
        if tag is None or tag != getattr(self, "_dd_tag", None):
            dd.set(D)
            self._dd_tag = tag
        return dd


def _is_tensor(x):
    try:
        import torch
        return isinstance(x, torch.Tensor)
    except ImportError:  # pragma: no cover
        return False


def _empty_mmap(shape):
    """Counterpart of lyssa.utils.dataset.get_empty_mmap (utils/dataset.py:136-149): float64 memmap on disk."""
    d = os.environ.get("LYSSA_MMAP_DIR", tempfile.gettempdir())
    fd, path = tempfile.mkstemp(prefix="lyssa_mmap_", suffix=".dat", dir=d)
    os.close(fd)
    return np.memmap(path, dtype=np.float64, mode='w+', shape=shape)


def batch_omp(X, Alpha, D, Gram, n_nonzero_coefs=None, tol=None):
    """Function form of lyssa/sparse_coding.py:302-367 for callers that precomputed Alpha and Gram.

    X and D are used for their shapes only (as in the reference, :306-307); ``tol`` is accepted and unused
    (as in the reference).  Alpha (n_atoms, n_samples), Gram (n_atoms, n_atoms) host arrays -> dense float64 Z.
    """
    import ctypes
    from . import _lib
    torch = engine.require_gpu()
    lib = _lib.load()
    K, N = int(D.shape[1]), int(X.shape[1])
    k = int(n_nonzero_coefs)
    Kp = _lib.padded_atoms(K)
    dev = engine.device_of(None)
    a0 = torch.zeros((N, Kp), dtype=torch.float32, device=dev)
    a0[:, :K] = torch.from_numpy(np.ascontiguousarray(np.asarray(Alpha, dtype=np.float32).T)).to(dev)
    G = torch.zeros((Kp, Kp), dtype=torch.float32, device=dev)
    G[:K, :K] = torch.from_numpy(np.asarray(Gram, dtype=np.float32)).to(dev)
    idx = torch.empty((N, k), dtype=torch.int32, device=dev)
    coef = torch.empty((N, k), dtype=torch.float32, device=dev)
    nnz = torch.empty((N,), dtype=torch.int32, device=dev)
    _lib.check(lib.lys_bomp_from_alpha0(ctypes.c_void_p(a0.data_ptr()), ctypes.c_void_p(G.data_ptr()), K, k, N,
                                        ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(coef.data_ptr()),
                                        ctypes.c_void_p(nnz.data_ptr()),
                                        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
               "lys_bomp_from_alpha0")
    return engine.densify(idx, coef, nnz, K)


def _thresh_from_alpha(Alpha, nonzero_percentage=None, n_nonzero_coefs=None):
    """`thresholding` / `soft_thresholding` (sparse_coding.py:416-425, feature_encoding.py:26-37) on a precomputed
    Alpha: the correlations are uploaded as the alpha0 tile and go through `thresh_wave_kernel` with an identity
    "dictionary" (alpha0 = Alpha'), i.e. the engine's own GEMM is bypassed, not the kernel."""
    Alpha = np.asarray(Alpha)
    K, N = Alpha.shape
    k = n_nonzero_coefs
    if nonzero_percentage is not None:
        k = int(np.floor(nonzero_percentage * K))
    # X := Alpha (features = atoms), D := I_K  =>  alpha0 = X D = Alpha'
    se = sparse_encoder(algorithm='thresh', params={'n_nonzero_coefs': int(k)}, verbose=False)
    return se.encode(Alpha, np.eye(K))


def thresholding(Alpha, nonzero_percentage=None, n_nonzero_coefs=None):
    """Function form of lyssa/sparse_coding.py:416-425."""
    return _thresh_from_alpha(Alpha, nonzero_percentage, n_nonzero_coefs)
