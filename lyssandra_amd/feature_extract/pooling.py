"""ScSPM spatial-pyramid pooling directly on the sparse codes.

Counterpart of the pooling loop of `sc_spm_extractor.encode` (lyssa/feature_extract/spatial_pyramid.py:57-97) with
`sc_max_pooling` (lyssa/feature_extract/pooling.py:4-7) and the optional `l2_normalizer`: the dense (K, n_patches)
code matrix is never built, the (atom, patch) non-zeros are max-reduced into the 1 + 4 + 16 cells with atomics.
"""
import ctypes

import numpy as np

from .. import _lib, engine


def pyramid_cells(pos, patch_size, imsize, levels=(1, 2, 4)):
    """Cell id of every patch at every level, offset so that ids index the flattened pyramid
    (spatial_pyramid.py:61-64,83-87): cy = py + psize/2 - 0.5, bin = floor(cy/hunit)*lev + floor(cx/wunit)."""
    pos = np.asarray(pos)
    py, px = pos[:, 0], pos[:, 1]
    cy = py + float(patch_size) / 2 - 0.5
    cx = px + float(patch_size) / 2 - 0.5
    cells = np.zeros((len(levels), pos.shape[0]), dtype=np.int32)
    off = 0
    for i, lev in enumerate(levels):
        wunit = float(imsize[1]) / lev
        hunit = float(imsize[0]) / lev
        b = np.floor(cy / hunit) * lev + np.floor(cx / wunit)
        ok = (b >= 0) & (b < lev * lev)            # the reference only visits j in range(lev^2)
        cells[i] = np.where(ok, b + off, -1).astype(np.int32)
        off += lev * lev
    return cells, off


def pool_cells_device(idx, coef, nnz, n_atoms, cells, n_cells, normalize=False):
    """max-|z| pooling of a device triplet into `n_cells` cells: cells [n_levels, N] int32 (host array or cuda tensor,
    -1 = not pooled at that level) -> cuda tensor [n_cells, n_atoms] fp32; `normalize` = l2-normalise every cell row."""
    torch = engine.require_gpu()
    lib = _lib.load()
    N, k = int(idx.shape[0]), int(idx.shape[1])
    cd = cells if isinstance(cells, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(cells))
    cd = cd.to(device=idx.device, dtype=torch.int32).contiguous()
    assert cd.shape[1] == N
    out = torch.empty((n_cells, n_atoms), dtype=torch.float32, device=idx.device)
    P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    _lib.check(lib.lys_pool_max_abs(P(idx), P(coef), P(nnz), k, N, P(cd), int(cd.shape[0]), n_atoms, n_cells, P(out),
                                    int(bool(normalize)), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
               "lys_pool_max_abs")
    return out


def spatial_pyramid_pool(idx, coef, nnz, n_atoms, pos, patch_size, imsize, levels=(1, 2, 4), normalize=False):
    """Device triplet of ONE image's patches -> flattened pyramid feature (n_cells * n_atoms,) float64,
    i.e. `poolpatches.flatten()` of spatial_pyramid.py:96."""
    cells, n_cells = pyramid_cells(pos, patch_size, imsize, levels)
    return pool_cells_device(idx, coef, nnz, n_atoms, cells, n_cells, normalize).double().cpu().numpy().reshape(-1)


class sc_max_pooling(object):
    """lyssa/feature_extract/pooling.py:4-7 (host callable kept for API parity; `sc_spm_extractor` recognises it and
    pools on the device instead)."""

    def __call__(self, Z):
        return np.max(np.abs(Z), axis=1)
