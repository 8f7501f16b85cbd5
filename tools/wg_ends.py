"""End times of every workgroup of a block-sweep launch (straggler hunt).  Needs the diagnostic build of ksvd_block.hip:
    BK_EXTRA=-DLYS_BK_WGEND bash tools/bk_dev_build.sh link
(lys_debug_timestamps then returns 64 + 520 values: the buffer below is sized for it)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lyssandra_amd import engine, _lib
n, K, k, N = 64, 1024, 10, 1 << 20
g = torch.Generator(device="cuda").manual_seed(3)
Xs = torch.randn((N, n), device="cuda", generator=g)
dd = engine.DeviceDictionary(n, K)
dd.set((Xs[:K] / Xs[:K].norm(dim=1, keepdim=True)).t().contiguous())
idx, coef, nnz = engine.bomp_encode(Xs, dd, k)
R, _ = engine.residual(Xs, dd, idx, coef, nnz, want_err=False)
ops = engine.HipBlockKsvdOps(R, dd, idx, coef, nnz, {})
ops.begin()
lib = _lib.load()
out = np.zeros(64 + 520 + 64, dtype=np.uint64)
def dump(tag):
    torch.cuda.synchronize()
    lib.lys_debug_timestamps(out.ctypes.data_as(ctypes.c_void_p))
    w = out[64:64 + 512].astype(np.int64).reshape(256, 2)
    st, en = w[:, 0], w[:, 1]
    b = st.min()
    st = (st - b) / 100.0; en = (en - b) / 100.0
    order = np.argsort(en)
    print(tag, "starts: max %.2f | ends: p50 %.2f p90 %.2f p99 %.2f max %.2f" % (st.max(), np.percentile(en, 50), np.percentile(en, 90), np.percentile(en, 99), en.max()))
    print("   latest 8 WGs (id:start..end):", " ".join("%d:%.1f..%.1f" % (i, st[i], en[i]) for i in order[-8:]))
    late = np.where(en > np.percentile(en, 90))[0]
    print("   WGs beyond p90:", late.tolist()[:40])
for c in range(0, 92):
    ops.step(0, c)
    if c in (40, 41, 90, 91): dump("X(%d)" % c)
    if c >= 1:
        ops.step(1, c)
        if c in (40, 41, 90, 91): dump("Y(%d)" % c)
