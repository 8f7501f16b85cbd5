"""Batching helpers and the process-map counterpart of the reference (lyssa/utils/__init__.py).

The reference parallelises by mapping column batches of the signal matrix over a multiprocessing Pool
(`run_parallel`, :40-163).  Here the unit of parallelism is the GPU: `shard_range` is the counterpart of
`gen_even_batches` for ranks, and the map itself is `lyssandra_amd.dist`.
"""
import multiprocessing

import numpy as np

from .math import fast_dot, outer, norm, norm_cols, normalize, frobenius_squared  # noqa: F401

cpu_count = multiprocessing.cpu_count()


def set_openblas_threads(n):
    """lyssa/utils/__init__.py:19-24 -- kept for call-site compatibility; the engine has no OpenBLAS pool."""
    return None


def get_openblas_threads():
    """lyssa/utils/__init__.py:27-30 -- the reference returns 0 when libopenblas was not found."""
    return 0


def gen_even_batches(N, n_batches):
    """lyssa/utils/__init__.py:166-180: n_batches-1 batches of floor(N/n_batches), the last takes the rest."""
    batch_size = int(np.floor(N / float(n_batches)))
    out, base = [], 0
    for _ in range(n_batches - 1):
        out.append(range(base, base + batch_size))
        base += batch_size
    out.append(range(base, N))
    return out


def gen_batches(N, batch_size=None):
    """lyssa/utils/__init__.py:183-201: fixed-size consecutive batches + remainder; None => a single batch."""
    if batch_size is None:
        return [range(0, N)]
    n_batches = int(np.floor(N / float(batch_size)))
    out, base = [], 0
    for _ in range(n_batches):
        out.append(range(base, base + batch_size))
        base += batch_size
    if N > base:
        out.append(range(base, N))
    return out


def run_parallel(func=None, data=None, args=None, batched_args=None,
                 result_shape=None, batch_size=None, n_batches=None,
                 mmap=False, msg=None, n_jobs=None):
    """Call-site compatible counterpart of lyssa/utils/__init__.py:40-163 for HOST functions.

    Same convention (first argument of `func` = the data batch, then the batched arguments, then the fixed
    ones) and the same batching: ``n_jobs == 1`` => one call on everything (:78-90); otherwise `n_batches` even
    column batches (`gen_even_batches`, the last one takes the remainder) or fixed-size batches, results scattered
    back into ``Z[:, idx]`` / ``Z[idx]`` (:143-146).  The batches run sequentially in this process: the engine's
    parallelism is the GPU, and forking a Pool after HIP initialisation is not safe (the reference already had to
    work around fork + OpenBLAS, sparse_coding.py:713-716).
    """
    is_array = isinstance(data, np.ndarray)
    n_samples = data.shape[1] if is_array else len(data)
    if args is None:
        args = ()
    Z = None
    result_is_array = False
    if result_shape is not None:
        if isinstance(result_shape, tuple):
            if mmap:
                from ..sparse_coding import _empty_mmap
                Z = _empty_mmap(result_shape)
            else:
                Z = np.zeros(result_shape)
            result_is_array = True
        else:
            Z = np.zeros(result_shape)
    if n_jobs == 1:
        _args = [data] + list(batched_args or []) + list(args)
        rs = func(*_args)
        if rs is not None:
            Z[:] = rs
        return Z
    if n_batches is not None:
        idx = gen_even_batches(n_samples, n_batches)
    else:
        idx = gen_batches(n_samples, batch_size=batch_size)
        n_batches = len(idx)

    def _slice(x, rng):
        if isinstance(x, np.ndarray):
            return x[:, rng.start:rng.stop]
        return x[rng.start:rng.stop]

    for i in range(n_batches):
        _args = [_slice(data, idx[i])] + [_slice(b, idx[i]) for b in (batched_args or [])] + list(args)
        rs = func(*_args)
        if rs is not None:
            if result_is_array:
                if rs.shape != Z[:, idx[i].start:idx[i].stop].shape:
                    raise ValueError("result of batch %d has shape %s, expected %s"
                                     % (i, rs.shape, Z[:, idx[i].start:idx[i].stop].shape))
                Z[:, idx[i].start:idx[i].stop] = rs
            else:
                Z[idx[i].start:idx[i].stop] = rs
    return Z


def shard_range(N, world_size, rank):
    """Contiguous signal range of `rank`: gen_even_batches(N, world_size)[rank] as (start, stop)."""
    size = N // world_size
    start = rank * size
    stop = N if rank == world_size - 1 else start + size
    return start, stop
