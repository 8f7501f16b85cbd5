"""Property tests of the CPU oracle / host logic (hypothesis): OMP invariants on random shapes, K-SVD sweep
monotonicity, online-DL statistics identities.  Small sizes: the whole CPU suite stays within minutes."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import lyssa_oracle as orc


def _problem(seed, n, K, N):
    rs = np.random.RandomState(seed)
    D = rs.randn(n, K)
    D /= np.sqrt((D * D).sum(0))
    return D, rs.randn(n, N)


@settings(max_examples=25, deadline=None)
@given(st.integers(0, 10 ** 6), st.integers(4, 24), st.integers(4, 40), st.integers(1, 6))
def test_omp_invariants(seed, n, K, k):
    k = min(k, n, K)
    D, X = _problem(seed, n, K, 6)
    idx, coef, nnz, gap = orc.bomp_encode_sparse(X, D, k)
    Z = orc.densify(idx, coef, nnz, K)
    assert np.array_equal(Z, orc.bomp_encode(X, D, k))
    for i in range(X.shape[1]):
        m = int(nnz[i])
        S = idx[i, :m]
        assert 1 <= m <= k and len(set(S.tolist())) == m
        ls = np.linalg.lstsq(D[:, S], X[:, i], rcond=None)[0]
        assert np.allclose(coef[i, :m], ls, atol=1e-8 * max(1.0, np.abs(ls).max()))   # coefficients = LS on the support
        r = X[:, i] - D[:, S] @ coef[i, :m]
        assert np.max(np.abs(D[:, S].T @ r)) < 1e-8 * max(1.0, np.linalg.norm(X[:, i]))  # residual _|_ support
        assert np.linalg.norm(r) <= np.linalg.norm(X[:, i]) + 1e-12
        first = int(np.argmax(np.abs(D.T @ X[:, i])))
        assert idx[i, 0] == first                                                      # greedy: best atom first


@settings(max_examples=10, deadline=None)
@given(st.integers(0, 10 ** 6))
def test_ksvd_sweep_never_increases_the_error_and_keeps_supports(seed):
    D, X = _problem(seed, 12, 20, 80)
    Z = orc.bomp_encode(X, D, 3)
    e0 = orc.approx_error(D, Z, X)
    supp = Z != 0
    D1, Z1, unused = orc.approx_ksvd(X, D.copy(), Z.copy(), n_cycles=1)
    assert orc.approx_error(D1, Z1, X) <= e0 * (1 + 1e-12)
    assert np.array_equal((Z1 != 0) | ~supp, np.ones_like(supp))        # support can only shrink by accident
    used = [a for a in range(20) if a not in unused]
    assert np.allclose(np.linalg.norm(D1[:, used], axis=0), 1.0, atol=1e-12)
    assert all(np.array_equal(D1[:, a], D[:, a]) for a in unused)


@settings(max_examples=10, deadline=None)
@given(st.integers(0, 10 ** 6), st.floats(0.0, 1.0))
def test_odl_statistics_identities(seed, beta):
    D, X = _problem(seed, 10, 16, 50)
    Z = orc.bomp_encode(X, D, 3)
    A0 = np.eye(16) * 0.5
    B0 = np.ones((10, 16)) * 0.1
    D1, A1, B1 = orc.odl_batch_update(D.copy(), A0, B0, X, Z, beta)
    assert np.allclose(A1, beta * A0 + Z @ Z.T) and np.allclose(B1, beta * B0 + X @ Z.T)
    assert np.allclose(A1, A1.T)
    assert np.allclose(np.linalg.norm(D1, axis=0), 1.0, atol=1e-12)
