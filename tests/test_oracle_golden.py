"""Pin the CPU oracle (oracle/lyssa_oracle.py) against outputs of the reference itself.

The golden vectors under tests/golden/ were produced by oracle/make_golden.py, which imports the
reference (py3-converted) in the build container.  Bar: identical supports, values within 1e-12.
"""
import numpy as np
import pytest

from oracle import lyssa_oracle as orc
from conftest import load_golden


def _triplet_sorted(Z, k):
    N = Z.shape[1]
    idx = -np.ones((N, k), dtype=np.int32)
    coef = np.zeros((N, k))
    nnz = np.zeros(N, dtype=np.int32)
    for i in range(N):
        nz = np.flatnonzero(Z[:, i])
        idx[i, :len(nz)] = nz
        coef[i, :len(nz)] = Z[nz, i]
        nnz[i] = len(nz)
    return idx, coef, nnz


@pytest.mark.parametrize("name", ["F1", "F2", "F3"])
def test_bomp_matches_reference(name):
    g = load_golden(name)
    X = g["X"].astype(np.float64).copy()
    D = g["D"].astype(np.float64).copy()
    k = int(g["k"])
    Z = orc.bomp_encode(X, D, k)
    idx, coef, nnz = _triplet_sorted(Z, k)
    assert np.array_equal(idx, g["idx"])
    assert np.array_equal(nnz, g["nnz"])
    assert np.max(np.abs(coef - g["coef"])) <= 1e-12
    # the sparse-triplet form of the oracle agrees with its own dense form
    si, sc, sn, gap = orc.bomp_encode_sparse(X, D, k)
    assert np.array_equal(orc.densify(si, sc, sn, D.shape[1]), Z)
    assert np.allclose(gap, g["gap"], rtol=0, atol=1e-9)


@pytest.mark.parametrize("case", ["dup", "exact2", "k1", "k4K4", "pool37", "nonunit", "ragged"])
def test_bomp_edge_cases_match_reference(case):
    g = load_golden("F4")
    X = g[case + "_X"].astype(np.float64)
    D = g[case + "_D"].astype(np.float64)
    k = int(g[case + "_k"])
    Z = orc.bomp_encode(X, D, k)
    Zr = g[case + "_Z"]
    assert Z.shape == Zr.shape
    if case == "exact2":
        # once the residual is ~1e-16 relative the selection is rounding noise:
        # compare the significant part only, the noise part must be tiny in both.
        big = np.abs(Zr) > 1e-10
        assert np.array_equal(np.abs(Z) > 1e-10, big)
        assert np.max(np.abs(Z[big] - Zr[big])) <= 1e-12
        assert np.max(np.abs(Z[~big])) < 1e-12 and np.max(np.abs(Zr[~big])) < 1e-12
        assert np.all(Z[:, -1] == 0)  # zero signal -> zero column
    else:
        assert np.array_equal(Z != 0, Zr != 0)
        assert np.max(np.abs(Z - Zr)) <= 1e-12


def test_approx_ksvd_matches_reference():
    g = load_golden("F5")
    X = g["X"].astype(np.float64)
    D = g["D0"].astype(np.float64).copy()
    k = int(g["k"])
    for it in range(3):
        Z = orc.bomp_encode(X, D, k)
        D, Z, unused = orc.approx_ksvd(X, D, Z, n_cycles=1)
        idx, coef, nnz = _triplet_sorted(Z, k)
        assert np.array_equal(idx, g["it%d_idx" % it])
        assert np.max(np.abs(coef - g["it%d_coef" % it])) <= 1e-10
        assert np.max(np.abs(D - g["it%d_D" % it])) <= 1e-10
        assert list(unused) == list(g["it%d_unused" % it])
        assert abs(orc.approx_error(D, Z, X) - float(g["it%d_err" % it])) <= 1e-8 * float(g["it%d_err" % it])
    D = g["D0"].astype(np.float64).copy()
    Z = orc.bomp_encode(X, D, k)
    D, Z, _ = orc.approx_ksvd(X, D, Z, n_cycles=2)
    assert np.max(np.abs(D - g["cyc2_D"])) <= 1e-10


def test_force_mi_matches_reference():
    """dict_learning/utils.py:86-139 on F11: same atoms replaced, same datapoints drawn (global RNG), for the oracle
    and for the product's host implementation (pure host control flow: no GPU needed)."""
    from lyssandra_amd.dict_learning.utils import force_mi as product_force_mi
    g = load_golden("F11")
    D0, X, Z = g["D"].astype(np.float64), g["X"].astype(np.float64), g["Z"]
    for fn in (orc.force_mi, product_force_mi):
        np.random.seed(4242)
        D1, un1 = fn(D0.copy(), X, Z, g["unused"].tolist(), float(g["eta"]))
        assert np.max(np.abs(D1 - g["D_out"])) <= 1e-12
        assert list(un1) == g["unused_out"].tolist()
        assert np.random.randint(0, 2 ** 31 - 1) == int(g["rng_after"])
    assert (np.abs(g["D_out"] - D0).max(0) > 0).sum() == 3          # the three coherent pairs each lost one atom


def test_lasso_oracle_against_independent_solvers():
    """'lasso' (sparse_coding.py:487-509) delegates to SPAMS, which is not available: the oracle restates the problem
    min 0.5||x-Da||^2 + lam||a||_1 and is pinned here against sklearn's coordinate-descent Lasso, sklearn's LARS path
    (SPAMS' own algorithm family) and the KKT conditions."""
    from sklearn.linear_model import Lasso, lars_path
    rs = np.random.RandomState(7)
    n, K, N = 32, 96, 12
    D = rs.randn(n, K)
    D /= np.linalg.norm(D, axis=0)
    X = rs.randn(n, N)
    X /= np.linalg.norm(X, axis=0)
    for lam in (0.4, 0.25):
        Z = orc.lasso_encode(X, D, lam)
        assert orc.lasso_kkt_violation(X, D, Z, lam) < 1e-10
        assert 0 < (Z != 0).sum(0).max() < n
        m = Lasso(alpha=lam / n, fit_intercept=False, tol=1e-14, max_iter=100000)
        for i in range(N):
            m.fit(D, X[:, i])
            assert np.max(np.abs(m.coef_ - Z[:, i])) < 1e-8
            _, _, coefs = lars_path(D, X[:, i], method='lasso', alpha_min=lam / n)
            assert np.max(np.abs(coefs[:, -1] - Z[:, i])) < 1e-8
    assert np.all(orc.lasso_encode(X, D, 1.0) == 0)          # lam >= max|D'x|: the zero code


def test_exact_ksvd_matches_reference():
    """ksvd.py:19-43 on F10 (reference run with randomized_svd seeded): the oracle's exact SVD reproduces the
    reference's atoms and codes up to the arbitrary sign of (d_k, x_k), and its error."""
    g5, g = load_golden("F5"), load_golden("F10")
    N, k = int(g["n_signals"]), int(g["k"])
    X = g5["X"].astype(np.float64)[:, :N]
    D = g5["D0"].astype(np.float64).copy()
    K = D.shape[1]
    for it in range(2):
        Z = orc.densify(g["it%d_idx" % it], g["it%d_coef_in" % it], g["it%d_nnz" % it], K)
        D1, Z1, unused = orc.ksvd_exact(X, D.copy(), Z.copy())
        Dr = g["it%d_D" % it]
        sgn = np.sign((D1 * Dr).sum(0))
        assert np.max(np.abs(D1 - Dr * sgn)) <= 1e-6
        Zr = orc.densify(g["it%d_idx" % it], g["it%d_coef" % it], g["it%d_nnz" % it], K)
        assert np.max(np.abs(Z1 - Zr * sgn[:, None])) <= 1e-6
        assert list(unused) == list(g["it%d_unused" % it])
        assert abs(orc.approx_error(D1, Z1, X) - float(g["it%d_err" % it])) <= 1e-8 * float(g["it%d_err" % it])
        D = Dr.copy()


@pytest.mark.parametrize("cycles", [0, 1, 3])
def test_nn_ksvd_matches_reference(cycles):
    """ksvd.py:46-95 on F14 (the reference's own seeded run): atoms, codes, unused atoms, the zero a clip creates.  The
    reference's solver returned u . d_old >= 0 for every atom of this fixture (recorded signs), i.e. the restatement's
    convention, so the comparison is direct."""
    g = load_golden("F14")
    X, D0, Z0 = g["X"].astype(np.float64), g["D0"].astype(np.float64), g["Z0"].astype(np.float64)
    signs = g["c%d_signs" % cycles]
    assert np.all(signs > 0)
    D1, Z1, unused = orc.nn_ksvd(X, D0.copy(), Z0.copy(), n_cycles=cycles, signs=signs)
    assert list(unused) == list(g["c%d_unused" % cycles])
    assert np.max(np.abs(D1 - g["c%d_D" % cycles])) <= 1e-9
    assert np.max(np.abs(Z1 - g["c%d_Z" % cycles])) <= 1e-9 * np.abs(Z1).max()
    assert np.array_equal(Z1 != 0, g["c%d_Z" % cycles] != 0)
    assert D1.min() >= 0 and Z1.min() >= 0
    assert np.allclose(np.linalg.norm(D1, axis=0), 1.0, atol=1e-12)


@pytest.mark.parametrize("verbose", [True, False])
def test_ksvd_driver_matches_reference(verbose):
    """Host control flow: patience quirk (11 encode calls for max_iter=50) and global-RNG consumption."""
    g = load_golden("F5")
    X = g["X"].astype(np.float64)[:, :600]
    k = int(g["k"])
    calls = []

    def enc(X_, D_):
        calls.append(1)
        return orc.bomp_encode(X_, D_, k)

    np.random.seed(1234)
    D, Z = orc.ksvd_dict_learn(X, 32, init_dict='data', encode=enc, max_iter=50, verbose=verbose)
    tag = "full_v%d" % int(verbose)
    assert len(calls) == int(g[tag + "_ncalls"]) == 11
    assert np.max(np.abs(D - g[tag + "_D"])) <= 1e-9
    assert np.random.randint(0, 2 ** 31 - 1) == int(g[tag + "_rng_after"])


def test_ksvd_driver_ndarray_init():
    g = load_golden("F5")
    X = g["X"].astype(np.float64)[:, :600]
    k = int(g["k"])
    D, Z = orc.ksvd_dict_learn(X, 128, init_dict=g["D0"].astype(np.float64),
                               encode=lambda X_, D_: orc.bomp_encode(X_, D_, k), max_iter=2, verbose=False)
    assert np.max(np.abs(D - g["init_nd_D"])) <= 1e-10
    idx, coef, nnz = _triplet_sorted(Z, k)
    assert np.array_equal(idx, g["init_nd_idx"])
    assert np.max(np.abs(coef - g["init_nd_coef"])) <= 1e-10


@pytest.mark.parametrize("tag,n_epochs,beta", [("e1", 1, None), ("e2", 2, None), ("e1b", 1, 0.9)])
def test_online_dict_learn_matches_reference(tag, n_epochs, beta):
    g5 = load_golden("F5")
    g = load_golden("F6")
    X = g5["X"].astype(np.float64)
    D0 = g5["D0"].astype(np.float64)
    k = int(g["k"])
    log = []

    def enc(X_, D_):
        log.append(np.array(D_, copy=True))
        return orc.bomp_encode(X_, D_, k)

    D, A, B = orc.online_dict_learn(X, D0.shape[1], encode=enc, batch_size=int(g["batch_size"]),
                                    D_init=D0.copy(), beta=beta, n_epochs=n_epochs)
    assert len(log) == int(g[tag + "_ncalls"])
    for i, Dl in enumerate(g[tag + "_Dlog"]):
        assert np.max(np.abs(log[i] - Dl)) <= 1e-10
    assert np.max(np.abs(D - g[tag + "_D"])) <= 1e-9
    assert np.max(np.abs(A - g[tag + "_A"])) <= 1e-9 * max(1.0, np.abs(g[tag + "_A"]).max())
    assert np.max(np.abs(B - g[tag + "_B"])) <= 1e-9 * max(1.0, np.abs(g[tag + "_B"]).max())


def test_batching_helpers():
    # lyssa/utils/__init__.py:166-201 (behaviour recorded in SURVEY appendix A)
    b = orc.gen_even_batches(37, 100)
    assert len(b) == 100 and all(len(x) == 0 for x in b[:99]) and list(b[99]) == list(range(37))
    b = orc.gen_even_batches(250, 100)
    assert [len(x) for x in b] == [2] * 99 + [52]
    assert [list(x) for x in orc.gen_batches(7, 3)] == [[0, 1, 2], [3, 4, 5], [6]]
    assert [list(x) for x in orc.gen_batches(6, 3)] == [[0, 1, 2], [3, 4, 5]]
    assert [list(x) for x in orc.gen_batches(5, None)] == [[0, 1, 2, 3, 4]]


def test_omp_invariants():
    """Algebraic properties of OMP (independent of the reference): |support|<=k, coefficients = least
    squares on the support, residual orthogonal to the selected atoms."""
    rs = np.random.RandomState(7)
    n, K, k, N = 24, 60, 5, 40
    D = rs.randn(n, K)
    D /= np.sqrt((D * D).sum(0))
    X = rs.randn(n, N)
    idx, coef, nnz, _ = orc.bomp_encode_sparse(X, D, k)
    for i in range(N):
        m = nnz[i]
        assert m == k and len(set(idx[i, :m])) == m
        S = idx[i, :m]
        ls = np.linalg.lstsq(D[:, S], X[:, i], rcond=None)[0]
        assert np.allclose(coef[i, :m], ls, atol=1e-10)
        r = X[:, i] - D[:, S] @ coef[i, :m]
        assert np.max(np.abs(D[:, S].T @ r)) < 1e-10


def test_omp_and_thresh_match_reference():
    """`_omp` (sparse_coding.py:19-57) and `thresholding` (:416-425) against the reference."""
    g = load_golden("F7")
    X, D, Dn = g["X"].astype(np.float64), g["D"].astype(np.float64), g["Dn"].astype(np.float64)
    for tag, DD in (("unit", D), ("nonunit", Dn)):
        Z = orc.omp_encode(X, DD, 6)
        assert np.array_equal(Z != 0, g["omp_%s_Z" % tag] != 0)
        assert np.max(np.abs(Z - g["omp_%s_Z" % tag])) <= 1e-11
    assert np.array_equal(orc.thresh_encode(X, D, n_nonzero_coefs=7), g["thresh_k7_Z"])
    assert np.array_equal(orc.thresh_encode(X, D, nonzero_percentage=0.4), g["thresh_p40_Z"])
    # on a unit-norm dictionary plain OMP and Batch-OMP are the same algorithm
    assert np.max(np.abs(orc.omp_encode(X, D, 6) - orc.bomp_encode(X, D, 6))) < 1e-6   # D is float32-rounded: diag = 1 +- 6e-8


def test_patches_preproc_pooling_match_reference():
    """grid_patches (utils/img.py:420-477), per-patch preproc (feature_extract/preproc.py:46-80)
    and the ScSPM pooling loop (feature_extract/spatial_pyramid.py:57-97) against the reference's outputs."""
    g = load_golden("F8")
    assert np.array_equal(orc.grid_patches(g["img_u8"], 8, 3), g["u8_p8_s3"])
    assert np.array_equal(orc.grid_patches(g["img_u8"], 16, 7), g["u8_p16_s7"])
    assert np.array_equal(orc.grid_patches(g["img_rgb"], 8, 5), g["rgb_p8_s5"])
    base = g["u8_p8_s3"].astype(np.float64)
    for name in ("scaling", "local_centering", "contrast_normalization", "normalization"):
        assert np.max(np.abs(orc.preproc(name, base) - g["pre_" + name])) <= 1e-13
    g9 = load_golden("F9")
    for tag, l2 in (("plain", False), ("l2", True)):
        f = orc.spm_pool(g9["Z"], g9["pos"], int(g9["patch_size"]), (int(g9["H"]), int(g9["W"])), l2=l2)
        assert np.max(np.abs(f - g9["feat_" + tag])) <= 1e-14


def test_c_oracle_matches_numpy_oracle_and_reference():
    """oracle/bomp_oracle.c (fast float64 C restatement used for the large GPU parity runs) against the numpy oracle
    and the reference-generated golden vectors, edge cases included."""
    from oracle import c_oracle
    for name in ("F1", "F2", "F3"):
        g = load_golden(name)
        X, D, k = g["X"].astype(np.float64), g["D"].astype(np.float64), int(g["k"])
        idx, coef, nnz, gap = c_oracle.bomp_encode_sparse(X, D, k)
        oi, oc, on, og = orc.bomp_encode_sparse(X, D, k)
        assert np.array_equal(idx, oi) and np.array_equal(nnz, on)
        assert np.max(np.abs(coef - oc)) <= 1e-12 and np.max(np.abs(gap - og)) <= 1e-9
        Z = orc.densify(idx, coef, nnz, D.shape[1])
        gi, gc, gn = _triplet_sorted(Z, k)
        assert np.array_equal(gi, g["idx"]) and np.max(np.abs(gc - g["coef"])) <= 1e-12
    g = load_golden("F4")
    for case in ("dup", "k1", "k4K4", "pool37", "nonunit", "ragged"):
        X, D, k = g[case + "_X"].astype(np.float64), g[case + "_D"].astype(np.float64), int(g[case + "_k"])
        Z = orc.densify(*c_oracle.bomp_encode_sparse(X, D, k)[:3], D.shape[1])
        assert np.array_equal(Z != 0, g[case + "_Z"] != 0) and np.max(np.abs(Z - g[case + "_Z"])) <= 1e-12, case


def test_c_oracle_approx_ksvd_matches_reference_and_numpy_oracle():
    """oracle/bomp_oracle.c::lyso_approx_ksvd (sparse-triplet float64 sweep used for the config-2-size GPU parity run)
    against the reference's own per-iteration outputs (F5) and the numpy oracle, unused atoms and n_cycles=2 included."""
    from oracle import c_oracle
    g = load_golden("F5")
    X = g["X"].astype(np.float64)
    k = int(g["k"])
    for it in range(3):
        Dprev = g["D0"].astype(np.float64) if it == 0 else g["it%d_D" % (it - 1)]
        gi, gc_in, gn = g["it%d_idx" % it], g["it%d_coef_in" % it], g["it%d_nnz" % it]
        D, coef, unused, err = c_oracle.approx_ksvd_sparse(X, Dprev, gi, gc_in, gn, n_cycles=1)
        assert np.max(np.abs(D - g["it%d_D" % it])) <= 1e-10
        assert np.max(np.abs(coef - g["it%d_coef" % it])) <= 1e-10
        assert unused == list(g["it%d_unused" % it])
        assert abs(err - float(g["it%d_err" % it])) <= 1e-8 * float(g["it%d_err" % it])
    D, _, _, _ = c_oracle.approx_ksvd_sparse(X, g["D0"].astype(np.float64), g["it0_idx"], g["it0_coef_in"],
                                             g["it0_nnz"], n_cycles=2)
    assert np.max(np.abs(D - g["cyc2_D"])) <= 1e-10
    # a shape with unused atoms, ragged supports and a coefficient that is exactly zero, against the numpy oracle
    rs = np.random.RandomState(5)
    n, K, kk, N = 24, 40, 4, 300
    Dm = orc.norm_cols(rs.randn(n, K))
    Xm = rs.randn(n, N)
    idx = -np.ones((N, kk), dtype=np.int32)
    coef = np.zeros((N, kk))
    nnz = rs.randint(0, kk + 1, size=N).astype(np.int32)
    for i in range(N):
        idx[i, :nnz[i]] = rs.choice(K - 3, size=nnz[i], replace=False)      # atoms K-3.. never used
        coef[i, :nnz[i]] = rs.randn(nnz[i])
    coef[7, 0] = 0.0                                                         # stored slot with a zero value: not in omega
    Z = orc.densify(idx, coef, nnz, K)
    Do, Zo, uo = orc.approx_ksvd(Xm, Dm.copy(), Z.copy(), n_cycles=2)
    Dc, cc, uc, err = c_oracle.approx_ksvd_sparse(Xm, Dm, idx, coef, nnz, n_cycles=2)
    assert uc == list(uo)
    assert np.max(np.abs(Dc - Do)) <= 1e-12
    assert np.max(np.abs(orc.densify(idx, cc, nnz, K) - Zo)) <= 1e-11
    assert abs(err - orc.approx_error(Do, Zo, Xm)) <= 1e-9 * err


def test_round2_widenings_match_reference():
    """F12 (generated by the reference itself): dataset-level preproc (feature_extract/preproc.py:18-31,55-62,77-78),
    error-constrained 'omp' (sparse_coding.py:27-31) and 'thresh' with 2048 atoms (:416-425) against the oracle."""
    g = load_golden("F12")
    Xp = g["pre_X"].astype(np.float64)
    for name in ("global_centering", "global_standarization", "whitening"):
        ref = g["pre_" + name]
        assert np.max(np.abs(orc.preproc(name, Xp) - ref)) <= 1e-10 * max(1.0, np.abs(ref).max()), name
    X = g["omp_X"].astype(np.float64)
    for tag, D in (("unit", g["omp_D"].astype(np.float64)), ("nonunit", g["omp_Dn"].astype(np.float64))):
        for tol in (2.0, 3.5):
            Z = orc.omp_encode(X, D, None, tol=tol)
            Zr = g["omp_%s_tol%g_Z" % (tag, tol)]
            assert np.array_equal(Z != 0, Zr != 0) and np.max(np.abs(Z - Zr)) <= 1e-10
            r = np.linalg.norm(X - D @ Z, axis=0)
            assert (r < tol).all()                                   # every signal stopped because ||r|| < tol
    Zt = orc.thresh_encode(g["th_X"].astype(np.float64), g["th_D"].astype(np.float64), n_nonzero_coefs=9)
    assert np.array_equal(Zt != 0, g["th_k9_Z"] != 0) and np.max(np.abs(Zt - g["th_k9_Z"])) <= 1e-12


def test_philox_generator_known_answers():
    """The synthetic-signal generator shared by bench.py's GPU and CPU legs: Philox4x32-10 against the
    known-answer vectors published with Random123, and the moments / shard consistency of the Gaussian stream."""
    import ctypes
    from oracle import c_oracle
    lib = c_oracle.load()
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        c = (ctypes.c_uint32 * 4)(*ctr)
        k = (ctypes.c_uint32 * 2)(*key)
        o = (ctypes.c_uint32 * 4)()
        lib.lyso_philox_block(c, k, o)
        assert tuple(o) == want
    X = c_oracle.synth_signals(11, 0, 100000, 64)
    assert abs(X.mean()) < 2e-3 and abs(X.std() - 1.0) < 2e-3 and abs((X.astype(np.float64) ** 4).mean() - 3.0) < 0.05
    # any shard regenerates the same values; a different seed / feature count gives a different / prefix-compatible stream
    assert np.array_equal(c_oracle.synth_signals(11, 7000, 300, 64), X[7000:7300])
    assert not np.array_equal(c_oracle.synth_signals(12, 0, 300, 64), X[:300])
    assert np.array_equal(c_oracle.synth_signals(11, 0, 300, 30), X[:300, :30])
