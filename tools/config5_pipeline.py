"""configs[4] at its real shape on synthetic data: ScSPM features of Caltech-101-sized image set (101 classes), then
LC-KSVD with one atom per training sample (30 per class: 3030 atoms, 3030 training columns, k = 30, stacked dimension
21 * 1024 + 3030 + 101 = 24 635) and prediction of the held-out images.  Prints per-stage wall times.

  python tools/config5_pipeline.py [n_classes=101] [imgs_per_class=35] [img_size=128] [lc_iters=2]
"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lyssandra_amd import engine
from lyssandra_amd.sparse_coding import sparse_encoder
from lyssandra_amd.feature_extract.spatial_pyramid import patch_extractor, sc_spm_extractor
from lyssandra_amd.feature_extract.pooling import sc_max_pooling
from lyssandra_amd.feature_extract.preproc import l2_normalizer
from lyssandra_amd.dict_learning.lc_ksvd import lc_ksvd_classifier
from lyssandra_amd.utils.math import norm_cols

n_classes = int(sys.argv[1]) if len(sys.argv) > 1 else 101
per_class = int(sys.argv[2]) if len(sys.argv) > 2 else 35
S = int(sys.argv[3]) if len(sys.argv) > 3 else 128
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 2
n_train, ps, step, K, k_patch = 30, 16, 8, 1024, 5
rs = np.random.RandomState(0)


def clock(label, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-44s %8.2f s" % (label, dt), flush=True)
    return r, dt


# class = a mixture of a few oriented gratings with class-specific frequencies + noise
yy, xx = np.meshgrid(np.arange(S), np.arange(S), indexing='ij')
imgs, labels = [], []
for c in range(n_classes):
    comps = [(rs.uniform(0.2, 1.2), rs.uniform(0, np.pi)) for _ in range(3)]
    for _ in range(per_class):
        g = sum(np.sin(f * (np.cos(a) * xx + np.sin(a) * yy) + rs.uniform(0, 2 * np.pi)) for f, a in comps)
        imgs.append((g + 0.3 * rs.randn(S, S)).astype(np.float32))
        labels.append(c)
labels = np.array(labels)
print("images: %d of %dx%d, %d classes" % (len(imgs), S, S, n_classes))
engine.require_gpu()
torch.zeros(1, device="cuda")

# patch dictionary: random unit-norm atoms (dictionary learning on patches is config 2's job)
Dp = rs.randn(ps * ps, K)
Dp /= np.linalg.norm(Dp, axis=0)
se_patch = sparse_encoder(algorithm='bomp', params={'n_nonzero_coefs': k_patch}, verbose=False)
ex = sc_spm_extractor(feature_extractor=patch_extractor(step_size=step, patch_size=ps), levels=(1, 2, 4),
                      sparse_coder=se_patch, pooling_operator=sc_max_pooling(), normalizer=l2_normalizer())
ex.encode(imgs[:4], Dp)                                  # warm-up (library load, first-use allocations)
F, t_feat = clock("ScSPM features (patches -> bomp -> pooling)", lambda: ex.encode(imgs, Dp))
n_patches = len(imgs) * ((S - ps) // step + 1) ** 2
print("   %d patches of %d dims, %.2f M patches/s end to end (image upload + feature download included); features %s"
      % (n_patches, ps * ps, n_patches / t_feat / 1e6, F.shape))
X = norm_cols(F)

np.random.seed(1)
se = sparse_encoder(algorithm='bomp', params={'n_nonzero_coefs': 30}, verbose=False)
lc = lc_ksvd_classifier(sparse_coder=se, max_iter=iters, n_class_samples=n_train, n_test_samples=None, n_tests=1,
                        param_grid=[{'alpha': [0.2], 'beta': [0.1]}])
import lyssandra_amd.dict_learning.lc_ksvd as lcm
_enc, _ksvd = se.__class__.__call__, lcm.ksvd
acc = {"encode": 0.0, "ksvd": 0.0}


def timed_ksvd(*a, **kw):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = _ksvd(*a, **kw)
    torch.cuda.synchronize(); acc["ksvd"] += time.perf_counter() - t0
    return r


def timed_enc(self, X_, D_):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = _enc(self, X_, D_)
    torch.cuda.synchronize(); acc["encode"] += time.perf_counter() - t0
    return r


lcm.ksvd = timed_ksvd
se.__class__.__call__ = timed_enc
_, t_lc = clock("lc_ksvd_classifier: split + %d LC-KSVD iterations + predict" % iters, lambda: lc(X, labels))
n_tr = n_train * n_classes
print("   stacked K-SVD problem: %d rows x %d training columns, %d atoms, k = 30" % (X.shape[0] + n_tr + n_classes, n_tr, n_tr))
print("   inside: sparse coding %.2f s (train x%d + test), exact K-SVD (host stacking, upload, %d atom updates/iter, download) %.2f s"
      % (acc["encode"], iters, n_tr, acc["ksvd"]))
print("   held-out accuracy %.3f (%d test images)" % (lc.best_score, len(imgs) - n_tr))
