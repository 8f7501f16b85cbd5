"""Where a step of bomp_block_kernel goes (configs[2] kernel shape: n = 256, K = 4096, k = 20), from the kernel's own
100 MHz wall-clock stamps.  Needs the development build of the library:

    python -c "from lyssandra_amd import build; build.build(True, extra_flags=['-DLYS_BLK_STAMPS'])"
    python tools/blk_timeline.py
    python -m lyssandra_amd.build        # back to the product build (bomp.hip is recompiled)

With LYS_ABL set every workgroup runs all k steps (exits disabled, pivots forced to 1: the results are meaningless, the
instruction stream and the memory traffic are those of a signal that needs all k atoms)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["LYS_ABL"] = "4096"
import numpy as np, torch
from lyssandra_amd import engine, _lib
lib = _lib.load()
n, K, k, N = 256, 4096, 20, 16384
g = torch.Generator(device="cuda").manual_seed(1)
Dt = torch.randn((n, K), device="cuda", generator=g); Dt = Dt / Dt.norm(dim=0, keepdim=True)
Xs = torch.randn((N, n), device="cuda", generator=g)
dd = engine.DeviceDictionary(n, K); dd.set(Dt)
for _ in range(3):
    engine.bomp_encode(Xs, dd, k)
torch.cuda.synchronize()
out = np.zeros(8 + 4 * 16384, dtype=np.uint64)
lib.lys_debug_blk_timeline.argtypes = [ctypes.c_void_p]
_lib.check(lib.lys_debug_blk_timeline(out.ctypes.data_as(ctypes.c_void_p)))
ph = out[:8].astype(np.float64) / (3 * N * k) / 100.0
print("us per step (thread 0 of every workgroup, mean): local argmax %.3f | barrier 1 %.3f | candidates + tests %.3f | Gram load + "
      "publish %.3f | barrier 2 %.3f | w, pivot %.3f | update + commit %.3f  -> %.3f" % (*ph[:7], ph[:7].sum()))
t = out[8:].reshape(-1, 4).astype(np.int64)
st, en, hw, steps = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
xcc = hw >> 32; hwid = hw & 0xffffffff
cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
key = xcc * 1000 + se * 100 + sh * 20 + cu
print("distinct (xcc,se,sh,cu):", len(np.unique(key)), " span of launch %.1f us" % ((en.max() - st.min()) / 100.0))
print("mean lifetime %.1f us, mean steps part %.1f us" % ((en - st).mean() / 100.0, steps.mean() / 100.0))
# concurrency per CU
mx = []
for kx in np.unique(key)[:8]:
    m = key == kx
    s, e = st[m], en[m]
    o = np.argsort(s)
    s, e = s[o], e[o]
    gaps = (s[1:] - e[:-1]) / 100.0
    ov = (s[1:] < e[:-1]).sum()
    print("cu key %d: %d WGs, overlapping successors %d, median gap %.2f us, lifetimes %.1f..%.1f" % (kx, m.sum(), ov, np.median(gaps), (e - s).min() / 100.0, (e - s).max() / 100.0))
print("WGs per cu key: min %d max %d" % (np.bincount(np.unique(key, return_inverse=True)[1]).min(), np.bincount(np.unique(key, return_inverse=True)[1]).max()))
