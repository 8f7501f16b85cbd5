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


def shard_range(N, world_size, rank):
    """Contiguous signal range of `rank`: gen_even_batches(N, world_size)[rank] as (start, stop)."""
    size = N // world_size
    start = rank * size
    stop = N if rank == world_size - 1 else start + size
    return start, stop
