"""Multi-GPU protocols of the hot path: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference's only parallel strategy is a data-parallel map over column batches of the signal matrix
(`run_parallel`, lyssa/utils/__init__.py:40-163: 100 even batches, results scattered back by column).  The GPU
counterpart: every rank owns ONE contiguous signal shard (`shard_range`, the `gen_even_batches` partition), the
dictionary is replicated.

  * encode           -- no collective at all (signals are independent given D);
  * approx K-SVD     -- exact reference semantics need, per atom IN ORDER, the sum over shards of the n+1 numbers
                        [sum_i R_i x_i , sum_i x_i^2] before the atom is normalised (lyssa/dict_learning/ksvd.py:116-119);
                        coefficient and residual updates are then shard-local (:121-123);
  * online DL        -- one all-reduce of [dA | dB] per mini-batch (online_dict_learn.py:84-85), then the identical
                        replicated update (:91-98);
  * errors, counts   -- scalar / K-vector all-reduces.

The protocol functions below are written against a small `ops` interface so that the SAME control flow runs on the
HIP engine (engine.HipKsvdOps / engine.OdlState) and, in the CPU tests (gloo, world_size 2), on a numpy stand-in.
"""
import os

import numpy as np

from .utils import shard_range  # noqa: F401  (re-export)


def _dist():
    import torch.distributed as dist
    return dist


def world(group=None):
    dist = _dist()
    if not dist.is_available() or not dist.is_initialized():
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


def allreduce_sum_(t, group=None):
    """In-place sum over ranks of a torch tensor (no-op without an initialised process group)."""
    dist = _dist()
    # LYS_DIST_FORCE=1: issue the collective also in a world of one rank (a self-reduction = identity) -- lets a single-GPU box
    # execute the RCCL branch below (tests/test_gpu_parity.py::test_rccl_backend_world_of_one)
    import os
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1
                                                          or os.environ.get("LYS_DIST_FORCE") == "1"):
        if not t.is_cuda and dist.get_backend(group) == "nccl":
            import torch
            d = t.to(torch.device("cuda", torch.cuda.current_device()))  # RCCL only moves device memory
            dist.all_reduce(d, op=dist.ReduceOp.SUM, group=group)
            t.copy_(d.cpu())
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def allreduce_max_(t, group=None):
    """In-place maximum over ranks of a torch tensor; like `allreduce_sum_` it stages a host tensor through the device under
    RCCL (which only moves device memory) and is a no-op without an initialised process group."""
    dist = _dist()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if not t.is_cuda and dist.get_backend(group) == "nccl":
            import torch
            d = t.to(torch.device("cuda", torch.cuda.current_device()))
            dist.all_reduce(d, op=dist.ReduceOp.MAX, group=group)
            t.copy_(d.cpu())
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t


def local_shard(X, group=None):
    """Columns of the (n, N) host matrix that belong to this rank (contiguous, remainder to the last rank)."""
    ws, rk = world(group)
    s, e = shard_range(X.shape[1], ws, rk)
    return X[:, s:e], (s, e)


def shard_minibatches(X, batch_size, group=None):
    """Columns of this rank such that LOCAL mini-batch b is this rank's shard of GLOBAL mini-batch b.

    The reference walks the signals in consecutive batches (`gen_batches`, lyssa/utils/__init__.py:183-201).  To
    keep exactly that sequence of (global) batches under data parallelism every rank takes the `shard_range` slice
    of EACH batch; running `online_dict_learn(..., batch_size=local_batch_size, group=...)` on the result then
    processes the same global batches as a single process would.  Returns (X_local, local_batch_size); when a batch does
    not split evenly over the ranks (the last rank takes the remainder, like `shard_range`) the second value is the LIST
    of this rank's local batch ranges instead of one size -- `online_dict_learn` accepts either as `batch_size`.
    """
    from .utils import gen_batches
    ws, rk = world(group)
    N = X.shape[1]
    cols, ranges, base, even = [], [], 0, True
    for b in gen_batches(N, batch_size):
        n_b = b.stop - b.start
        even = even and (n_b % ws == 0)
        s, e = shard_range(n_b, ws, rk)
        cols.append(np.arange(b.start + s, b.start + e))
        ranges.append(range(base, base + (e - s)))
        base += e - s
    idx = np.concatenate(cols) if cols else np.zeros(0, dtype=int)
    if batch_size is None:
        return X[:, idx], None
    if even and ranges:
        return X[:, idx], len(ranges[0])
    return X[:, idx], ranges


# ------------------------------------------------------------------------------------------------ approx K-SVD
def ksvd_cycle_sharded(ops, K, group=None):
    """One dictionary-update cycle over signal shards.  `ops` provides, for the LOCAL shard:

        ops.local_counts()        -> int64 tensor [K]: number of local non-zeros per atom
        ops.accumulate(a)         -> enqueue phase 1 of atom a into ops.stats(a)
        ops.stats(a)              -> fp64 tensor [n+1] (view into the statistics buffer) to be all-reduced
        ops.apply(a)              -> phase 2 of atom a from the reduced statistics (also publishes d_new)
        ops.fused_step(a)         -> optional: apply(a-1) + accumulate(a) in one launch, a in [0, K]
        ops.commit(global_counts) -> D[a] <- d_new for every atom that is used on ANY rank

    Returns the list of atoms unused on every rank (ksvd.py:111-115).  Atoms are visited strictly in order.
    """
    counts = ops.local_counts()
    allreduce_sum_(counts, group)
    if getattr(ops, "has_fused", hasattr(ops, "fused_step")):
        # one launch per atom: [pending update of a-1] + [accumulation for a], then the all-reduce of stats(a)
        for a in range(K + 1):
            ops.fused_step(a)
            if a < K:
                allreduce_sum_(ops.stats(a), group)
    else:
        for a in range(K):
            ops.accumulate(a)
            allreduce_sum_(ops.stats(a), group)
            ops.apply(a)
    ops.commit(counts)
    return [int(a) for a in (counts == 0).nonzero().flatten().tolist()]


def ksvd_exact_cycle_sharded(ops, K, group=None):
    """One cycle of the EXACT rank-1 update (ksvd.py:19-43) over signal shards.  The sufficient statistic of an atom is the
    n x n Gram matrix of its restricted residual: per atom IN ORDER

        ops.gram(a)   -> fp64 tensor [n, n]: this shard's Rk Rk'           (all-reduced here)
        ops.update(a) -> replicated eigen-solve on the reduced matrix + local coefficient / residual update

    plus `ops.local_counts()`, `ops.set_used(global_counts)` and `ops.commit()`.  K collectives of n^2 doubles per cycle
    (32 KB at n = 64): latency-bound, like the reference semantics demand (atom a+1 reads the residual atom a left).
    Returns the atoms unused on every rank."""
    if getattr(ops, "matrix_free", False):           # n > 256: no Gram matrix to exchange
        return ksvd_exact_cycle_sharded_mf(ops, K, group)
    counts = ops.local_counts()
    allreduce_sum_(counts, group)
    ops.set_used(counts)
    host_counts = counts.cpu().tolist()
    for a in range(K):
        if host_counts[a] == 0:
            continue                                   # unused everywhere: keeps its column (ksvd.py:27-29)
        allreduce_sum_(ops.gram(a), group)
        ops.update(a)
    ops.commit()
    return [a for a in range(K) if host_counts[a] == 0]


def ksvd_exact_cycle_sharded_mf(ops, K, group=None, max_it=400, poll=4, sin2_tol=1e-12):
    """One cycle of the EXACT rank-1 update (ksvd.py:19-43) over signal shards when n is large (n > 256 on the device:
    LC-KSVD's stacked signals, lc_ksvd.py:165).  Neither Gram matrix can be exchanged there -- n x n is 4.8 GB at n = 24 635
    and the column Gram matrix couples the shards -- so the leading pair comes from a matrix-free power iteration on
    Rk Rk' started at d_old, with ONE all-reduce of an n-vector per iteration.  Per atom IN ORDER

        ops.mf_begin(a)        -> u = d_old                                                  (replicated)
        repeat:  ops.mf_iterate(a) -> fp32 / fp64 tensor [n]: this shard's sum_i (rk_i . u / ||u||) rk_i   (all-reduced here)
                 ops.mf_norm(a)    -> u = the reduced vector; norms and the angle to the previous iterate    (replicated)
                 every `poll` iterations: stop when ops.mf_sin2() <= sin2_tol (successive iterates within 1e-6 rad) -- the
                 number is computed from the reduced vector, so every rank takes the same decision
        ops.mf_apply(a)        -> x_i = rk_i . u, R_i = rk_i - u x_i on the local rows; the new atom on every rank

    plus `local_counts`, `set_used`, `commit` like ksvd_exact_cycle_sharded.  Returns the atoms unused on every rank."""
    counts = ops.local_counts()
    allreduce_sum_(counts, group)
    ops.set_used(counts)
    host_counts = counts.cpu().tolist()
    for a in range(K):
        if host_counts[a] == 0:
            continue
        ops.mf_begin(a)
        for it in range(int(max_it)):
            allreduce_sum_(ops.mf_iterate(a), group)
            ops.mf_norm(a)
            if (it + 1) % int(poll) == 0 and not (ops.mf_sin2() > sin2_tol):
                break
        ops.mf_apply(a)
    ops.commit()
    return [a for a in range(K) if host_counts[a] == 0]


def nn_ksvd_cycle_sharded(ops, K, n_cycles, group=None):
    """One pass of the NON-NEGATIVE K-SVD update (`nn_ksvd`, ksvd.py:46-95) over signal shards (round 4).  Per atom IN ORDER:
    the Gram matrix of the restricted residual is all-reduced like for the exact update (replicated rank-1 solve), then every
    projection pass exchanges the ONE quantity it needs from the other shards --

        ops.gram(a)          -> fp64 [n, n]  this shard's Rk Rk'                                   (all-reduced here)
        ops.nn_begin(a)      -> replicated eigen-solve u; local x = max(Rk'u, 0)
        ops.nn_scalar()      -> fp64 [1]: the LOCAL x'x of the pass just run                        (all-reduced here)
        ops.nn_project(a)    -> replicated: d = max(u, 0), the skip test of ksvd.py:79-82 on the global x'x
        n_cycles times:
          ops.nn_accumulate(a) ; ops.nn_vector() -> fp64 [n]: local Rk x                            (all-reduced here)
          ops.nn_step(a)       -> replicated d = max(s / x'x, 0); local x = max(Rk'd / d'd, 0)     (then nn_scalar again)
        ops.nn_commit(a)     -> d /= ||d||, x *= ||d||, residual rows (local); the new atom on every rank

    plus `local_counts`, `set_used`, `commit` like the exact update.  2 + 2 n_cycles small collectives per atom on top of the
    Gram matrix; a skipped atom still takes part in them (zeros), so no rank waits for another.  Returns the unused atoms."""
    counts = ops.local_counts()
    allreduce_sum_(counts, group)
    ops.set_used(counts)
    host_counts = counts.cpu().tolist()
    for a in range(K):
        if host_counts[a] == 0:
            continue
        allreduce_sum_(ops.gram(a), group)
        ops.nn_begin(a)
        allreduce_sum_(ops.nn_scalar(), group)
        ops.nn_project(a)
        for _ in range(int(n_cycles)):
            ops.nn_accumulate(a)
            allreduce_sum_(ops.nn_vector(), group)
            ops.nn_step(a)
            allreduce_sum_(ops.nn_scalar(), group)
        ops.nn_commit(a)
    ops.commit()
    return [a for a in range(K) if host_counts[a] == 0]


def ksvd_cycle_blocks(ops, group=None):
    """One cycle of the BLOCK sweep over signal shards (csrc/ksvd_block.hip).  Per block c of B atoms:

        ops.step(0, c) -> X(c): the B sequential atom updates of block c-1 from its (reduced) statistics slab,
                                replicated on every rank, next to block c's statistics over the local signals that do
                                not use block c-1
        ops.step(1, c) -> Y(c): block c-1 applied to the local residual rows / codes + the rest of block c's statistics
        ops.slab(c)    -> fp64 tensor [stride]: per-atom sums (ksvd.py:118), counts and tuple moments of block c

    i.e. K/B all-reduces of one slab each instead of K all-reduces of n+1 numbers, all enqueued on the stream without
    a host synchronisation (the collective is asynchronous with respect to the host).  Returns the atoms unused on
    every rank (the counts travel inside the slabs)."""
    ops.begin()
    for c in range(ops.nb + 1):
        ops.step(0, c)
        if c >= 1:
            ops.step(1, c)
        if c < ops.nb:
            allreduce_sum_(ops.slab(c), group)
    return ops.finish()


# ------------------------------------------------------------------------------------------------ online DL
_SYM_FLAT = {}  # persistent exchange buffer of allreduce_symmetric_ (device path)


def allreduce_symmetric_(A, extra=None, group=None, block=1024):
    """Sum over ranks of a SYMMETRIC square matrix `A` (and optionally of `extra`, any tensor) with one collective that
    carries only the block-upper triangle of A: the row blocks [i*block, (i+1)*block) x [i*block, K) are packed into one
    flat buffer together with `extra`, all-reduced, unpacked and mirrored.  For Z Z' of online DL at K = 8192 this is
    144 MB + B instead of 256 MB + B on the wire (online_dict_learn.py:84-85).  No-op without a process group."""
    import torch
    ws, _ = world(group)
    if ws <= 1 and os.environ.get("LYS_DIST_FORCE") != "1":     # (forced: the pack / collective / unpack run in a world of one)
        return A
    K = A.shape[0]
    assert A.shape[0] == A.shape[1]
    if A.is_cuda and A.dtype == torch.float32 and A.is_contiguous() and K % 64 == 0 and block % 64 == 0 and \
            (extra is None or (extra.is_cuda and extra.dtype == torch.float32 and extra.is_contiguous())):
        # device path: ONE persistent flat buffer [packed upper blocks | extra], written by lys_sym_pack and one copy,
        # all-reduced, read back by lys_sym_unpack (scatter + LDS-tiled mirror) and one copy -- no torch.cat, no per-block
        # slicing, no strided transposes
        import ctypes
        from . import _lib
        lib = _lib.load()
        npk = int(lib.lys_sym_packed_count(K, block))
        nex = 0 if extra is None else extra.numel()
        key = (A.device, K, block, nex)
        flat = _SYM_FLAT.get(key)
        if flat is None:
            _SYM_FLAT.clear()                      # one shape at a time (K = 8192: 148 MB)
            flat = _SYM_FLAT[key] = torch.empty((npk + nex,), dtype=torch.float32, device=A.device)
        st = ctypes.c_void_p(torch.cuda.current_stream(A.device).cuda_stream)
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        _lib.check(lib.lys_sym_pack(P(A), K, block, P(flat), st), "lys_sym_pack")
        if nex:
            flat[npk:].copy_(extra.reshape(-1))
        allreduce_sum_(flat, group)
        _lib.check(lib.lys_sym_unpack(P(flat), K, block, P(A), st), "lys_sym_unpack")
        if nex:
            extra.copy_(flat[npk:].view_as(extra))
        return A
    parts = [A[i:min(i + block, K), i:].reshape(-1) for i in range(0, K, block)]
    if extra is not None:
        parts.append(extra.reshape(-1))
    flat = torch.cat(parts)
    allreduce_sum_(flat, group)
    off = 0
    for i in range(0, K, block):
        h = min(i + block, K) - i
        blk = flat[off:off + h * (K - i)].view(h, K - i)
        off += h * (K - i)
        A[i:i + h, i:] = blk
        if i + h < K:
            A[i + h:, i:i + h] = blk[:, h:].t()
    if extra is not None:
        extra.copy_(flat[off:].view_as(extra))
    return A


def odl_batch_sharded(ops, beta, non_neg=False, group=None):
    """One mini-batch of online DL over shards: local increments, ONE all-reduce of [upper(dA) | dB], replicated update.

        ops.increments() -> (dA, dB) tensors holding the LOCAL Z Z' (symmetric) and X Z'
        ops.update(beta, non_neg) -> A = beta A + dA; B = beta B + dB; dictionary update
    """
    dA, dB = ops.increments()
    allreduce_symmetric_(dA, extra=dB, group=group)
    ops.update(beta, non_neg)


# ------------------------------------------------------------------------------------------------ host helpers
def init_dictionary_sharded(X_local, span, N_total, n_atoms, group=None):
    """`init_dictionary(method='data')` (lyssa/dict_learning/utils.py:49-70) when the signals are sharded.

    Every rank must hold the same numpy global-RNG state (seed identically): the candidate list (columns with
    energy > 1e-6) is all-gathered, the SAME `np.random.choice` draw is made everywhere, chosen columns are
    contributed by their owners through one all-reduce.  Returns (D (n, K) float64, unused_data as GLOBAL indices).
    """
    import torch
    from .utils.math import norm_cols
    n = X_local.shape[0]
    s, e = span
    mask = torch.zeros((N_total,), dtype=torch.int32)
    mask[s:e] = torch.from_numpy((np.einsum('ij,ij->j', X_local, X_local) > 1e-6).astype(np.int32))
    allreduce_sum_(mask, group)
    idxs = np.flatnonzero(mask.numpy()).tolist()
    if len(idxs) < n_atoms:
        raise ValueError("not enough datapoints to initialize the dictionary")
    subset = np.random.choice(len(idxs), size=n_atoms, replace=False)
    subset_idxs = np.array(idxs).astype(int)[subset]
    D = torch.zeros((n, n_atoms), dtype=torch.float64)
    for j, g in enumerate(subset_idxs):
        if s <= g < e:
            D[:, j] = torch.from_numpy(np.asarray(X_local[:, g - s], dtype=np.float64))
    allreduce_sum_(D, group)
    D = norm_cols(D.numpy().copy())
    chosen = set(subset_idxs.tolist())
    return D, [x for x in idxs if x not in chosen]


def fetch_global_column(X_local, span, g, group=None):
    """Column `g` (global index) of the sharded signal matrix on every rank (owner contributes, others add zeros)."""
    import torch
    s, e = span
    v = torch.zeros((X_local.shape[0],), dtype=torch.float64)
    if s <= g < e:
        v += torch.from_numpy(np.asarray(X_local[:, g - s], dtype=np.float64))
    allreduce_sum_(v, group)
    return v.numpy()


def force_mi_sharded(D, X_local, Z_local, span, unused_data, eta, group=None, max_tries=100):
    """`force_mi` (lyssa/dict_learning/utils.py:86-139, the `eta` step of ksvd_dict_learn, ksvd.py:209-213) when the signals
    and codes are sharded: a replicated host decision.  It needs |D'D| (D is replicated), the code-row norms ||Z[k, :]|| --
    ONE all-reduce of the K local sums of squares -- and the candidate datapoints by GLOBAL index (`fetch_global_column`,
    the owner contributes).  Every rank holds the same numpy RNG state and therefore draws the same candidates.
    `Z_local`: dense (K, N_local) host codes or the device triplet."""
    import torch
    from .dict_learning.utils import _code_row_norms, force_mi
    K = np.asarray(D).shape[1]
    sq = torch.from_numpy(np.asarray(_code_row_norms(Z_local, K), dtype=np.float64) ** 2)
    allreduce_sum_(sq, group)
    return force_mi(D, None, None, unused_data, eta, max_tries=max_tries,
                    fetch_column=lambda g: fetch_global_column(X_local, span, g, group), usage=np.sqrt(sq.numpy()))
