"""ScSPM feature extraction (lyssa/feature_extract/spatial_pyramid.py) as ONE device pipeline -- config 5's front end.

The reference loops over the images: `grid_patches` -> `sparse_coder.encode` (a dense (K, n_patches) float64 matrix per
image) -> per-cell `np.nonzero` / pooling / normalising (spatial_pyramid.py:45-97).  Here the patches of a whole chunk of
images are produced on the device (`lys_grid_patches`, signal-major fp32), encoded in ONE launch sequence (alpha0 GEMM +
greedy kernel), and the sparse triplet is max-|z|-pooled straight into the (image, cell) rows of the feature matrix
(`lys_pool_max_abs`); neither the host patch matrix nor the dense codes exist.
"""
import numpy as np

from .. import engine
from ..utils.img import compute_n_patches, grid_patches_device
from .pooling import pyramid_cells, pool_cells_device, sc_max_pooling
from .preproc import l2_normalizer

_CHUNK_PATCHES = 1 << 22       # patches encoded per launch sequence (bounds the alpha0 tile workspace, not the result)


class patch_extractor(object):
    """lyssa/feature_extract/spatial_pyramid.py:24-33.  `extract` -> (patches (dim, n) host array, pos (n, 2) top-left
    (row, col) of every patch, row-major over the grid); `extract_device` -> the same patches as a signal-major cuda
    tensor.  (The reference's `grid_patches` ignores its `scale` / `return_loc` arguments, utils/img.py:427-489, so its
    `extract` cannot unpack the result; the evident intent -- raw patches plus grid positions -- is what this returns.)"""

    def __init__(self, step_size=None, patch_size=None):
        self.step_size = step_size
        self.patch_size = patch_size

    def positions(self, imshape):
        n_h, n_w = compute_n_patches(imshape[0], imshape[1], self.patch_size, self.step_size)
        ys, xs = np.meshgrid(np.arange(n_h) * self.step_size, np.arange(n_w) * self.step_size, indexing='ij')
        return np.stack([ys.ravel(), xs.ravel()], axis=1)

    def extract_device(self, img, device=None):
        img = np.asarray(img)
        return grid_patches_device(img, self.patch_size, self.step_size, device=device), self.positions(img.shape)

    def extract(self, img):
        Xs, pos = self.extract_device(img)
        return Xs.t().contiguous().double().cpu().numpy(), pos


class sc_spm_extractor(object):
    """lyssa/feature_extract/spatial_pyramid.py:36-97: `encode(imgs, dictionary)` -> (sum(levels^2) * n_atoms, n_imgs)."""

    def __init__(self, feature_extractor=None, levels=(1, 2, 4), sparse_coder=None, pooling_operator=None, normalizer=None):
        self.feature_extractor = feature_extractor
        self.levels = levels
        self.sparse_coder = sparse_coder
        self.pooling_operator = pooling_operator
        self.normalizer = normalizer

    def _device_plan(self):
        if self.pooling_operator is not None and not isinstance(self.pooling_operator, sc_max_pooling):
            raise NotImplementedError("only sc_max_pooling (max |z| per cell, ScSPM) is pooled on the device")
        if self.normalizer is not None and not isinstance(self.normalizer, l2_normalizer):
            raise NotImplementedError("only l2_normalizer (or None) is applied on the device")
        return self.normalizer is not None

    def encode(self, imgs, dictionary):
        torch = engine.require_gpu()
        normalize = self._device_plan()
        psize = self.feature_extractor.patch_size
        n_atoms = int(dictionary.shape[1])
        n_cells = int(np.sum(np.array(self.levels) ** 2))
        dd = self.sparse_coder._dictionary(dictionary)
        Z = np.zeros((n_cells * n_atoms, len(imgs)))
        start = 0
        while start < len(imgs):
            # one chunk of images: patches + cell ids (offset by the image's slot in the chunk)
            tiles, cells, total = [], [], 0
            stop = start
            while stop < len(imgs) and (stop == start or total < _CHUNK_PATCHES):
                img = np.asarray(imgs[stop])
                if hasattr(self.feature_extractor, "extract_device"):
                    Xs, pos = self.feature_extractor.extract_device(img, dd.device)
                else:
                    desc, pos = self.feature_extractor.extract(img)
                    Xs = engine.signals_to_device(desc, dd.device)
                c, _ = pyramid_cells(pos, psize, img.shape, self.levels)
                cells.append(np.where(c >= 0, c + (stop - start) * n_cells, -1))
                tiles.append(Xs)
                total += int(Xs.shape[0])
                stop += 1
            Xs = tiles[0] if len(tiles) == 1 else torch.cat(tiles, dim=0)
            idx, coef, nnz = self.sparse_coder.encode_device(Xs, dd)
            pooled = pool_cells_device(idx, coef, nnz, n_atoms, np.concatenate(cells, axis=1).astype(np.int32),
                                       (stop - start) * n_cells, normalize)
            Z[:, start:stop] = pooled.view(stop - start, n_cells * n_atoms).t().double().cpu().numpy()
            start = stop
        return Z


def pyramid_feat_extract(imgs, extractor, D):
    return extractor.encode(imgs, D)


class spatial_pyramid(object):
    """lyssa/feature_extract/spatial_pyramid.py:104-149 without the workspace store (storage is out of scope): the
    dictionary is set on `.D` or learned by `dict_learn`, `extract` returns the ScSPM feature matrix."""

    def __init__(self, mmap=False, workspace=None, metadata=None):
        self.workspace = workspace
        self.metadata = metadata
        self.D = None
        self.mmap = mmap

    def extract(self, imgs, pyramid_feat_extractor=None, save=False, n_jobs=1):
        if self.D is None:
            raise ValueError("spatial_pyramid.extract: no dictionary (set .D or call dict_learn)")
        return pyramid_feat_extractor.encode(imgs, self.D)

    def dict_learn(self, imgs, feature_extractor=None, dict_learner=None):
        self.descriptors = feature_extractor(imgs)
        dict_learner.fit(self.descriptors)
        self.D = dict_learner.D
