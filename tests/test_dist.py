"""N > 1 path on CPU: world_size-2 gloo processes run the SAME protocol code as the GPU engine
(lyssandra_amd.dist) with a numpy stand-in for the shard-local kernels, and must reproduce the single-process
oracle (lyssa/dict_learning/ksvd.py:98-126, online_dict_learn.py:84-98) on the full data."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

EPS = np.finfo(np.float64).eps


class NumpyKsvdOps(object):
    """Shard-local phases of one approx-K-SVD atom update in numpy (test stand-in for engine.HipKsvdOps)."""

    def __init__(self, Y, D, Z):
        self.D, self.Z = D, Z
        self.R = Y - D @ Z
        n, K = D.shape
        self.sbuf = torch.zeros((K, n + 1), dtype=torch.float64)
        self.Dnext = np.zeros_like(D)

    def local_counts(self):
        return torch.from_numpy((self.Z != 0).sum(axis=1).astype(np.int64))

    def accumulate(self, a):
        om = self.Z[a] != 0
        x = self.Z[a, om]
        self.sbuf[a, :-1] = torch.from_numpy(self.R[:, om] @ x)
        self.sbuf[a, -1] = float(np.dot(x, x))

    def stats(self, a):
        return self.sbuf[a]

    def apply(self, a):
        s = self.sbuf[a].numpy()
        v = s[:-1] + self.D[:, a] * s[-1]
        dn = v / (np.sqrt(np.dot(v, v)) + EPS)
        self.Dnext[:, a] = dn
        om = self.Z[a] != 0
        if om.any():
            xo = self.Z[a, om]
            xn = self.R[:, om].T @ dn + xo * float(self.D[:, a] @ dn)
            self.R[:, om] += np.outer(self.D[:, a], xo) - np.outer(dn, xn)
            self.Z[a, om] = xn

    def commit(self, counts):
        used = counts.numpy() > 0
        self.D[:, used] = self.Dnext[:, used]


class NumpyExactKsvdOps(object):
    """Shard-local phases of the exact rank-1 update in numpy (stand-in for engine.HipExactKsvdOps): the n x n Gram
    matrix of the restricted residual is all-reduced, the eigen-solve is replicated."""

    def __init__(self, Y, D, Z):
        self.D, self.Z = D, Z
        self.R = Y - D @ Z
        self.C = torch.zeros((D.shape[0], D.shape[0]), dtype=torch.float64)
        self.Dnext = D.copy()
        self.used = None

    def local_counts(self):
        return torch.from_numpy((self.Z != 0).sum(axis=1).astype(np.int64))

    def set_used(self, counts):
        self.used = counts.numpy() > 0

    def gram(self, a):
        om = self.Z[a] != 0
        Rk = self.R[:, om] + np.outer(self.D[:, a], self.Z[a, om])
        self.C[:] = torch.from_numpy(Rk @ Rk.T)
        return self.C

    def update(self, a):
        w, V = np.linalg.eigh(self.C.numpy())
        u = V[:, -1]
        if np.dot(u, self.D[:, a]) < 0:
            u = -u
        self.Dnext[:, a] = u
        om = self.Z[a] != 0
        if om.any():
            Rk = self.R[:, om] + np.outer(self.D[:, a], self.Z[a, om])
            x = Rk.T @ u
            self.R[:, om] = Rk - np.outer(u, x)
            self.Z[a, om] = x

    def commit(self):
        self.D[:, self.used] = self.Dnext[:, self.used]


class NumpyExactMfOps(NumpyExactKsvdOps):
    """Shard-local phases of the matrix-free exact update (dist.ksvd_exact_cycle_sharded_mf) in numpy: stand-in for
    engine.HipExactKsvdOps at n > 256 (lys_ksvd_exact_mf_phase)."""
    matrix_free = True

    def _rk(self, a):
        om = self.Z[a] != 0
        return om, self.R[:, om] + np.outer(self.D[:, a], self.Z[a, om])

    def mf_begin(self, a):
        self.u = self.D[:, a].copy()
        self.sin2 = 1.0
        self.un = torch.zeros((self.D.shape[0],), dtype=torch.float64)

    def mf_iterate(self, a):
        om, Rk = self._rk(a)
        v = Rk.T @ (self.u / np.linalg.norm(self.u))
        self.un[:] = torch.from_numpy(Rk @ v)
        return self.un

    def mf_norm(self, a):
        un = self.un.numpy().copy()
        den = (un @ un) * (self.u @ self.u)
        self.sin2 = max(0.0, 1.0 - (un @ self.u) ** 2 / den) if den > 0 else 0.0
        self.u = un

    def mf_sin2(self):
        return self.sin2

    def mf_apply(self, a):
        nrm = np.linalg.norm(self.u)
        u = self.u / nrm if nrm > 0 else self.D[:, a].copy()
        if np.dot(u, self.D[:, a]) < 0:
            u = -u
        self.Dnext[:, a] = u
        om, Rk = self._rk(a)
        if om.any():
            x = Rk.T @ u
            self.R[:, om] = Rk - np.outer(u, x)
            self.Z[a, om] = x


class NumpyNnKsvdOps(NumpyExactKsvdOps):
    """Shard-local phases of nn_ksvd (ksvd.py:46-95) in numpy: stand-in for engine.HipNnKsvdOps."""

    def __init__(self, Y, D, Z):
        NumpyExactKsvdOps.__init__(self, Y, D, Z)
        self.sc = torch.zeros((1,), dtype=torch.float64)
        self.vec = torch.zeros((D.shape[0],), dtype=torch.float64)
        self.eps = float(np.finfo(np.float64).eps)

    def _rk(self, a):
        om = self.Z[a] != 0
        return om, self.R[:, om] + np.outer(self.D[:, a], self.Z[a, om])

    def nn_begin(self, a):
        w, V = np.linalg.eigh(self.C.numpy())
        u = V[:, -1]
        if np.dot(u, self.D[:, a]) < 0:
            u = -u
        self.u = u
        self.om, self.Rk = self._rk(a)
        self.x = np.maximum(self.Rk.T @ u, 0.0)
        self.sc[0] = float(self.x @ self.x)

    def nn_scalar(self):
        return self.sc

    def nn_vector(self):
        return self.vec

    def nn_project(self, a):
        self.d = np.maximum(self.u, 0.0)
        self.xtx = float(self.sc[0])
        self.skip = (self.d @ self.d <= self.eps) or (self.xtx <= self.eps)

    def nn_accumulate(self, a):
        self.xtx = float(self.sc[0])          # the all-reduced x'x of the pass that produced self.x
        self.vec[:] = torch.from_numpy(np.zeros_like(self.d) if self.skip else self.Rk @ self.x)

    def nn_step(self, a):
        if self.skip:
            self.sc[0] = 0.0
            return
        self.d = np.maximum(self.vec.numpy() / self.xtx, 0.0)
        self.x = np.maximum(self.d @ self.Rk / (self.d @ self.d), 0.0)
        self.sc[0] = float(self.x @ self.x)

    def nn_commit(self, a):
        if self.skip:
            return
        nrm = np.sqrt(self.d @ self.d)
        self.Dnext[:, a] = self.d / nrm
        self.Z[a, self.om] = self.x * nrm
        self.R[:, self.om] = self.Rk - np.outer(self.Dnext[:, a], self.Z[a, self.om])

    def commit(self):
        self.D[:, self.used] = self.Dnext[:, self.used]


class NumpyBlockKsvdOps(object):
    """float64 numpy stand-in for engine.HipBlockKsvdOps: the block Gauss-Seidel sweep of csrc/ksvd_block.hip (B atoms
    per step, exact in-block coupling through tuple moments) with the same `ops` interface and the same slab layout
    [S (B x (n+2)) | Q (G x n) | C (G x B) | GC (G)], G = 2^B - 1 - B, g = (2^t - 1 - t) + (pi - 1)."""

    def __init__(self, Y, D, Z, B):
        self.D, self.Z, self.B = D, Z, B
        self.R = Y - D @ Z
        self.n, self.K = D.shape
        self.nb = (self.K + B - 1) // B
        self.G = (1 << B) - 1 - B
        n, G = self.n, self.G
        self.offQ = B * (n + 2)
        self.offC = self.offQ + G * n
        self.offGC = self.offC + G * B
        self.stride = self.offGC + G
        self.stats = torch.zeros((self.nb, self.stride), dtype=torch.float64)
        self.Dnext = D.copy()

    def begin(self):
        self.stats.zero_()
        self.support = self.Z != 0            # omega of this cycle (ksvd.py:111)

    def slab(self, c):
        return self.stats[c]

    def _in_block(self, i, c):
        lo, hi = c * self.B, min(self.K, (c + 1) * self.B)
        return [a for a in range(lo, hi) if self.support[a, i]]

    def _accumulate(self, c, with_prev):
        n, B = self.n, self.B
        s = self.stats[c].numpy()
        for i in range(self.Z.shape[1]):
            atoms = self._in_block(i, c)
            if not atoms:
                continue
            uses_prev = c >= 1 and bool(self._in_block(i, c - 1))
            if uses_prev != with_prev:
                continue
            r = self.R[:, i]
            pi = 0
            for j, a in enumerate(atoms):
                t, x = a - c * B, self.Z[a, i]
                if j == 0:
                    s[t * (n + 2):t * (n + 2) + n] += x * r
                else:
                    g = ((1 << t) - 1 - t) + pi - 1
                    s[self.offQ + g * n:self.offQ + (g + 1) * n] += x * r
                    for b in atoms[:j]:
                        s[self.offC + g * B + (b - c * B)] += x * self.Z[b, i]
                    s[self.offGC + g] += 1
                s[t * (n + 2) + n] += x * x
                s[t * (n + 2) + n + 1] += 1
                pi |= 1 << t

    def _narrow(self, c):
        n, B = self.n, self.B
        s = self.stats[c].numpy()
        for t in range(B):
            a = c * B + t
            if a >= self.K:
                break
            if s[t * (n + 2) + n + 1] == 0:
                continue                       # unused atom keeps its column
            v = s[t * (n + 2):t * (n + 2) + n] + self.D[:, a] * s[t * (n + 2) + n]
            for pi in range(1, 1 << t):
                g = ((1 << t) - 1 - t) + pi - 1
                if s[self.offGC + g] == 0:
                    continue
                u = s[self.offQ + g * n:self.offQ + (g + 1) * n].copy()
                for l in range(t):
                    if (pi >> l) & 1:
                        u += s[self.offC + g * B + l] * self.D[:, c * B + l]
                        dn = self.Dnext[:, c * B + l]
                        u -= dn * np.dot(dn, u)
                v += u
            self.Dnext[:, a] = v / (np.sqrt(np.dot(v, v)) + EPS)

    def _apply(self, p):
        for i in range(self.Z.shape[1]):
            for a in self._in_block(i, p):     # ascending: the sequential order of ksvd.py:105-123
                xo = self.Z[a, i]
                v = self.R[:, i] + self.D[:, a] * xo
                xn = float(np.dot(v, self.Dnext[:, a]))
                self.R[:, i] = v - self.Dnext[:, a] * xn
                self.Z[a, i] = xn

    def step(self, mode, c):
        if mode == 0:
            if c >= 1:
                self._narrow(c - 1)
            if c < self.nb:
                self._accumulate(c, with_prev=False)
        else:
            self._apply(c - 1)
            if c < self.nb:
                self._accumulate(c, with_prev=True)

    def finish(self):
        n = self.n
        cnt = self.stats[:, :self.B * (n + 2)].reshape(self.nb, self.B, n + 2)[:, :, n + 1].reshape(-1)[:self.K]
        self.D[:] = self.Dnext
        return [int(a) for a in (cnt == 0).nonzero().flatten().tolist()]


class NumpyOdlOps(object):
    def __init__(self, D, A, B, Xb, Zb):
        self.D, self.A, self.B, self.Xb, self.Zb = D, A, B, Xb, Zb

    def increments(self):
        self.dA = torch.from_numpy(self.Zb @ self.Zb.T)
        self.dB = torch.from_numpy(self.Xb @ self.Zb.T)
        return self.dA, self.dB

    def update(self, beta, non_neg=False):
        self.A[:] = beta * self.A + self.dA.numpy()
        self.B[:] = beta * self.B + self.dB.numpy()
        DA = self.D @ self.A
        for k in range(self.D.shape[1]):
            self.D[:, k] = (1 / (self.A[k, k] + EPS)) * (self.B[:, k] - DA[:, k]) + self.D[:, k]
        if non_neg:
            self.D[self.D < 0] = 0
        self.D /= (np.sqrt((self.D * self.D).sum(0)) + EPS)[None, :]


def _worker(rank, world, port, out):  # noqa: C901
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lyssandra_amd import dist as ld
        from oracle import lyssa_oracle as orc
        rs = np.random.RandomState(5)
        n, K, k, N = 16, 24, 3, 203          # N not divisible by world: remainder goes to the last rank
        D0 = rs.randn(n, K)
        D0 /= np.sqrt((D0 * D0).sum(0))
        X = rs.randn(n, N)
        X[:, 7] = 0.0                         # a zero-energy column is never an init candidate
        Z = orc.bomp_encode(X, D0, k)
        Z[K - 1, :] = 0.0                     # make the last atom unused on every rank
        Xl, span = ld.local_shard(X)
        assert span == ld.shard_range(N, world, rank)
        # ---- approx K-SVD: sharded protocol == sequential reference semantics
        Dl, Zl = D0.copy(), Z[:, span[0]:span[1]].copy()
        ops = NumpyKsvdOps(Xl, Dl, Zl)
        unused = ld.ksvd_cycle_sharded(ops, K)
        Dref, Zref = D0.copy(), Z.copy()
        Dref, Zref, unused_ref = orc.approx_ksvd(X, Dref, Zref, n_cycles=1)
        assert unused == list(unused_ref) == [K - 1]
        assert np.max(np.abs(Dl - Dref)) < 1e-12
        assert np.max(np.abs(Zl - Zref[:, span[0]:span[1]])) < 1e-12
        # ---- exact rank-1 update: one Gram-matrix all-reduce per atom == the exact SVD on the full data
        Dl, Zl = D0.copy(), Z[:, span[0]:span[1]].copy()
        unused = ld.ksvd_exact_cycle_sharded(NumpyExactKsvdOps(Xl, Dl, Zl), K)
        De, Ze, ue = orc.ksvd_exact(X, D0.copy(), Z.copy())
        assert unused == list(ue) == [K - 1]
        assert np.max(np.abs(Dl - De)) < 1e-9, np.max(np.abs(Dl - De))
        assert np.max(np.abs(Zl - Ze[:, span[0]:span[1]])) < 1e-9
        # ---- the same update without a Gram matrix (n > 256 on the device): power iteration, one n-vector all-reduce per
        # iteration, stopped at 1e-6 rad between successive iterates == the exact SVD to the iteration's accuracy
        Dl, Zl = D0.copy(), Z[:, span[0]:span[1]].copy()
        unused = ld.ksvd_exact_cycle_sharded(NumpyExactMfOps(Xl, Dl, Zl), K)
        assert unused == list(ue) == [K - 1]
        assert np.max(np.abs(Dl - De)) < 2e-5, np.max(np.abs(Dl - De))
        assert np.max(np.abs(Zl - Ze[:, span[0]:span[1]])) < 2e-5 * np.abs(Ze).max()
        # ---- nn_ksvd on shards: Gram matrix + one scalar / one n-vector per projection pass == the oracle on the full data
        rsn = np.random.RandomState(11)
        Dn = np.abs(rsn.randn(n, K)) + 0.05
        Dn /= np.sqrt((Dn * Dn).sum(0))
        Zn = np.zeros((K, N))
        for i in range(N):
            Zn[rsn.choice(K - 1, k, replace=False), i] = np.abs(rsn.randn(k)) + 0.1
        Xn = np.abs(Dn @ Zn + 0.05 * rsn.randn(n, N))
        users = np.flatnonzero(Zn[0] != 0)
        Xn[:, users] -= np.outer(Dn[:, 0], 3.0 * Zn[0, users])      # atom 0: x = max(Rk'u, 0) = 0 -> skipped (ksvd.py:79-82)
        for cyc in (0, 2):
            Dl, Zl = Dn.copy(), Zn[:, span[0]:span[1]].copy()
            unused = ld.nn_ksvd_cycle_sharded(NumpyNnKsvdOps(Xn[:, span[0]:span[1]], Dl, Zl), K, cyc)
            Dq, Zq, uq = orc.nn_ksvd(Xn, Dn.copy(), Zn.copy(), n_cycles=cyc)
            assert unused == list(uq) == [K - 1]
            assert np.array_equal(Dq[:, 0], Dn[:, 0]) and np.array_equal(Dl[:, 0], Dn[:, 0])     # the skipped atom
            assert np.max(np.abs(Dl - Dq)) < 1e-9, (cyc, np.max(np.abs(Dl - Dq)))
            assert np.max(np.abs(Zl - Zq[:, span[0]:span[1]])) < 1e-9
        # ---- the block sweep (one slab all-reduce per block of B atoms) == sequential reference semantics
        for B in (4, 8):
            Dl, Zl = D0.copy(), Z[:, span[0]:span[1]].copy()
            unused = ld.ksvd_cycle_blocks(NumpyBlockKsvdOps(Xl, Dl, Zl, B))
            assert unused == [K - 1]
            assert np.max(np.abs(Dl - Dref)) < 1e-11, (B, np.max(np.abs(Dl - Dref)))
            assert np.max(np.abs(Zl - Zref[:, span[0]:span[1]])) < 1e-11
        # ---- online DL: one all-reduce of [dA | dB] per batch == full-batch statistics
        Do, Ao, Bo = D0.copy(), np.zeros((K, K)), np.zeros((n, K))
        Dg, Ag, Bg = D0.copy(), np.zeros((K, K)), np.zeros((n, K))
        for beta in (0.0, 0.7):
            Zb = orc.bomp_encode(X, Dg, k)
            ld.odl_batch_sharded(NumpyOdlOps(Do, Ao, Bo, Xl, orc.bomp_encode(Xl, Do, k)), beta)
            Dg, Ag, Bg = orc.odl_batch_update(Dg, Ag, Bg, X, Zb, beta)
            assert np.max(np.abs(Ao - Ag)) < 1e-10 and np.max(np.abs(Bo - Bg)) < 1e-10
            assert np.max(np.abs(Do - Dg)) < 1e-10
        # ---- sharded init_dictionary == the reference's on the full matrix (same global RNG state)
        np.random.seed(77)
        Di, unused_data = ld.init_dictionary_sharded(Xl, span, N, 9)
        np.random.seed(77)
        Dfull, unused_full = orc.init_dictionary(X.copy(), 9, return_unused_data=True)
        assert np.max(np.abs(Di - Dfull)) < 1e-14 and unused_data == unused_full and 7 not in unused_data
        col = ld.fetch_global_column(Xl, span, 150)
        assert np.array_equal(col, X[:, 150])
        # ---- eta / force_mi on shards (ksvd.py:209-213): a replicated decision from all-reduced code-row norms and globally
        # fetched candidate columns == the reference's force_mi on the full matrices (same global RNG state)
        Dc = D0.copy()
        for a_, b_ in ((2, 11), (5, 20)):                       # two strongly coherent pairs
            v = Dc[:, a_] + 0.05 * rs.randn(n)
            Dc[:, b_] = v / np.sqrt((v * v).sum())
        pool = [i for i in range(40, 120)]
        np.random.seed(99)
        Dm, un_m = ld.force_mi_sharded(Dc.copy(), Xl, Z[:, span[0]:span[1]], span, list(pool), 0.9)
        rng_after = np.random.randint(0, 2 ** 31 - 1)
        np.random.seed(99)
        Dmf, un_f = orc.force_mi(Dc.copy(), X, Z, list(pool), 0.9)
        assert np.random.randint(0, 2 ** 31 - 1) == rng_after                  # same number of draws
        assert un_m == un_f and len(un_f) < len(pool)
        assert np.max(np.abs(Dm - Dmf)) < 1e-14 and np.abs(Dm - Dc).max() > 0
        # ---- mini-batch sharding: local batch b == this rank's slice of global batch b
        Xm = np.arange(2 * 20.0).reshape(2, 20)
        Xloc, lbs = ld.shard_minibatches(Xm, 8)                 # global batches 8, 8, 4 -> local 4, 4, 2
        assert lbs == 4 and Xloc.shape == (2, 10)
        exp = [c for b0, nb in ((0, 8), (8, 8), (16, 4)) for c in range(b0 + rank * nb // 2, b0 + (rank + 1) * nb // 2)]
        assert Xloc[0].tolist() == [float(c) for c in exp]
        # uneven batches (7, 7, 6 over 2 ranks: 3|4, 3|4, 3|3): explicit local ranges, remainder to the last rank
        Xloc, lranges = ld.shard_minibatches(Xm, 7)
        assert [len(r) for r in lranges] == ([3, 3, 3] if rank == 0 else [4, 4, 3])
        exp = [c for b0, nb in ((0, 7), (7, 7), (14, 6)) for c in range(b0 + (rank * nb) // 2, b0 + ((rank + 1) * nb) // 2)]
        assert Xloc[0].tolist() == [float(c) for c in exp] and lranges[-1].stop == Xloc.shape[1]
        # ---- a batch with FEWER signals than ranks (the remainder batch of 9 signals in batches of 8): rank 0's local
        # range is empty, it contributes zero statistics and still joins the all-reduce (a rank that skipped the
        # collective, or raised on N = 0, would hang the others)
        Xm1 = rs.randn(n, 9)
        Xloc1, lr1 = ld.shard_minibatches(Xm1, 8)
        assert [len(r) for r in lr1] == ([4, 0] if rank == 0 else [4, 1])
        Xlast = Xm1[:, 8:9]                                      # the one-signal global batch
        Xl1 = Xloc1[:, lr1[1].start:lr1[1].stop]
        assert Xl1.shape[1] == (0 if rank == 0 else 1)
        Do1, Ao1, Bo1 = D0.copy(), np.zeros((K, K)), np.zeros((n, K))
        Zl1 = orc.bomp_encode(Xl1, Do1, k) if Xl1.shape[1] else np.zeros((K, 0))
        ld.odl_batch_sharded(NumpyOdlOps(Do1, Ao1, Bo1, Xl1, Zl1), 0.0)
        Dg1, Ag1, Bg1 = orc.odl_batch_update(D0.copy(), np.zeros((K, K)), np.zeros((n, K)), Xlast,
                                             orc.bomp_encode(Xlast, D0, k), 0.0)
        assert np.max(np.abs(Ao1 - Ag1)) < 1e-12 and np.max(np.abs(Bo1 - Bg1)) < 1e-12 and np.max(np.abs(Do1 - Dg1)) < 1e-10
        # ---- scalar reduction used for the error
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        ld.allreduce_sum_(t)
        assert t.item() == world * (world + 1) / 2
        # ---- maximum over ranks (lc_ksvd(group=): the number of classes when a shard does not hold every label)
        tm = torch.tensor([3 + 2 * rank], dtype=torch.int64)
        ld.allreduce_max_(tm)
        assert tm.item() == 3 + 2 * (world - 1)
        out[rank] = 1
    finally:
        dist.destroy_process_group()


def test_sharded_protocols_gloo_world2():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert dict(out) == {0: 1, 1: 1}


@pytest.mark.parametrize("B", [4, 8])
def test_block_sweep_algebra_matches_sequential_oracle(B):
    """The block Gauss-Seidel sweep (csrc/ksvd_block.hip) is the reference's sequential sweep, not an approximation:
    float64 numpy restatement of the X/Y step protocol vs oracle approx_ksvd (ksvd.py:98-126), on codes dense enough
    that most signals use 2-4 atoms of a block (K = 21 atoms, k = 6), two cycles, K not a multiple of B, one unused atom."""
    from lyssandra_amd import dist as ld
    from oracle import lyssa_oracle as orc
    rs = np.random.RandomState(11 + B)
    n, K, k, N = 12, 21, 6, 150
    D0 = orc.norm_cols(rs.randn(n, K))
    X = rs.randn(n, N)
    Z = orc.bomp_encode(X, D0, k)
    Z[5, :] = 0.0
    Dl, Zl = D0.copy(), Z.copy()
    ops = NumpyBlockKsvdOps(X, Dl, Zl, B)
    u1 = ld.ksvd_cycle_blocks(ops)
    u2 = ld.ksvd_cycle_blocks(ops)
    Dref, Zref, uref = orc.approx_ksvd(X, D0.copy(), Z.copy(), n_cycles=2)
    assert u1 + u2 == list(uref) == [5, 5]
    assert np.max(np.abs(Dl - Dref)) < 1e-11 and np.max(np.abs(Zl - Zref)) < 1e-11
    assert np.max(np.abs(ops.R - (X - Dl @ Zl))) < 1e-11


def test_single_process_is_a_noop_group():
    from lyssandra_amd import dist as ld
    t = torch.ones(3)
    assert ld.world() == (1, 0) and torch.equal(ld.allreduce_sum_(t), torch.ones(3))
    X = np.arange(12.0).reshape(2, 6)
    Xl, span = ld.local_shard(X)
    assert span == (0, 6) and Xl.shape == (2, 6)
