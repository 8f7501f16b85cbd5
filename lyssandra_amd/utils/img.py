"""Patch extraction feeding the encode path -- counterpart of lyssa/utils/img.py:258-489.

`grid_patches_device` is the fast path: the image is uploaded once and the patches are produced ON THE DEVICE, in the
engine's signal-major fp32 layout, with the per-patch preprocessing fused -- no host float64 (n, N) matrix, no
transpose, no second pass.  `grid_patches` / `extract_patches` keep the reference's host-array signatures.
"""
import ctypes

import numpy as np

from .. import _lib, engine


def compute_n_patches(h, w, patch_size, step_size, padding=False):
    """lyssa/utils/img.py:258-273."""
    pad = int(np.floor(patch_size / 2)) if padding else 0
    return ((h - patch_size + pad) // step_size) + 1, ((w - patch_size + pad) // step_size) + 1


def grid_patches_device(img, patch_size, step_size, scale=1.0, center=False, normalize=False, device=None):
    """Image (H, W) or (H, W, C), uint8 or float -> signal-major fp32 cuda tensor [n_patches, patch_size^2 * C].

    scale / center / normalize = the per-patch steps of lyssa/feature_extract/preproc.py ('scaling' => scale=1/255,
    'local_centering' => center, 'contrast_normalization' => center + normalize, 'normalization' => normalize)."""
    torch = engine.require_gpu()
    lib = _lib.load()
    dev = engine.device_of(device)
    img = np.asarray(img)
    if img.ndim == 2:
        img = img[:, :, None]
    if img.ndim != 3:
        raise ValueError('image must be a 2D or 3D np.array')
    H, W, C = img.shape
    if img.dtype == np.uint8:
        t, dtype = torch.from_numpy(np.ascontiguousarray(img)).to(dev), 0
    else:
        t, dtype = torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32)).to(dev), 1
    n_ph, n_pw = compute_n_patches(H, W, patch_size, step_size)
    dim = patch_size * patch_size * C
    Xs = torch.empty((n_ph * n_pw, dim), dtype=torch.float32, device=dev)
    _lib.check(lib.lys_grid_patches(ctypes.c_void_p(t.data_ptr()), dtype, H, W, C, patch_size, step_size, float(scale),
                                    int(bool(center)), int(bool(normalize)), ctypes.c_void_p(Xs.data_ptr()),
                                    engine._ld(Xs), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
               "lys_grid_patches")
    return Xs


def grid_patches(img, patch_size=None, step_size=None, n_patches=None, return_loc=False, scale=False):
    """lyssa/utils/img.py:420-489 -> host array (patch_size^2 * C, n_patches), the input's dtype.

    ``n_patches`` (random subset, step 1) draws from the GLOBAL numpy RNG exactly like the reference (:480-485)."""
    img = np.asarray(img)
    n_req = n_patches
    if n_req is not None:
        step_size = 1
    Xs = grid_patches_device(img, patch_size, step_size)
    P = Xs.t().contiguous().cpu().numpy().astype(img.dtype if img.dtype != np.float32 else np.float32)
    if n_req is not None and n_req < P.shape[1]:
        sel = np.random.choice(np.arange(P.shape[1]), n_req, replace=False)
        P = P[:, sel]
    return P


def extract_patches(imgs, step_size=None, n_patches=None, patch_size=None, mmap=False, scale=False, verbose=False,
                    mem="high", n_jobs=1):
    """lyssa/utils/img.py:300-376: patches of a list of images, concatenated along the columns.

    Returns ``(patches, patch_numbers)`` when ``step_size`` is given, else ``patches`` (like the reference)."""
    n_imgs = len(imgs)
    per_img = None
    if n_patches is not None:
        per_img = int(np.floor(float(n_patches) / float(n_imgs)))
    cols, numbers = [], []
    for im in imgs:
        if per_img is not None:
            P = grid_patches(im, patch_size=patch_size, n_patches=per_img, scale=scale)
        else:
            P = grid_patches(im, patch_size=patch_size, step_size=step_size, scale=scale)
        cols.append(np.asarray(P, dtype=np.float64))
        numbers.append(P.shape[1])
    patches = np.concatenate(cols, axis=1) if cols else np.zeros((0, 0))
    if step_size is not None:
        return patches, np.array(numbers).astype(int)
    return patches
