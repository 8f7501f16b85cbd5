"""Phase timestamps of the block K-SVD sweep kernels (debug hook lys_debug_timestamps), config-2 shape."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lyssandra_amd import engine, _lib
n, K, k, N = 64, 1024, 10, (int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20)
g = torch.Generator(device="cuda").manual_seed(3)
Xs = torch.randn((N, n), device="cuda", generator=g)
dd = engine.DeviceDictionary(n, K)
dd.set((Xs[:K] / Xs[:K].norm(dim=1, keepdim=True)).t().contiguous())
idx, coef, nnz = engine.bomp_encode(Xs, dd, k)
R, _ = engine.residual(Xs, dd, idx, coef, nnz, want_err=False)
ops = engine.HipBlockKsvdOps(R, dd, idx, coef, nnz, {})
ops.begin()
lib = _lib.load()
out = np.zeros(64, dtype=np.uint64)


def show(tag, names):
    torch.cuda.synchronize()
    lib.lys_debug_timestamps(out.ctypes.data_as(ctypes.c_void_p))
    w = out.astype(np.int64)
    for wg, o in (("wg0", 32), ("wg/2", 48)):
        t = [(w[o + i] - w[o]) / 100.0 if w[o + i] else None for i in range(len(names))]
        t += [(w[o + i] - w[o]) / 100.0 if w[o + i] else None for i in (6, 7)]
        names = list(names) + ["fast loop done", "barrier 1 passed"]
        print("%s %-4s " % (tag, wg) + " | ".join("%s %s" % (nm, ("%.2f" % v) if v is not None else "-") for nm, v in zip(names, t)))
    t = (w[:6] - w[0]) / 100.0
    if w[60] and w[61] and w[32]:
        print("   all workgroups: last start %.2f, last end %.2f us after wg0's start" % ((w[60] - w[32]) / 100.0, (w[61] - w[32]) / 100.0))
    if w[0] and w[32] and w[48]:  # absolute skew of the three stamped workgroups (100-MHz counter is global)
        base = min(w[0], w[32], w[48])
        print("   absolute (us from the earliest start): narrow %.2f..%.2f | wg0 %.2f..%.2f | wg/2 %.2f..%.2f"
              % ((w[0] - base) / 100.0, (w[5] - base) / 100.0, (w[32] - base) / 100.0, (w[37] - base) / 100.0,
                 (w[48] - base) / 100.0, (w[53] - base) / 100.0))
    if w[12] > 0:  # -DLYS_BK_ATOM_STAMPS build: core-clock cycles per phase, mean over the block's atoms 1..B-1 (team 0)
        print("   atom loop (cycles per atom, %d atoms): evaluate %.0f | rendezvous %.0f | slot sum + norm %.0f | rsq + d_new %.0f"
              % (w[12], w[8] / w[12], w[9] / w[12], w[10] / w[12], w[11] / w[12]))
    return t


show_at = set(int(v) for v in os.environ.get("BK_SHOW", "1,2,3,4").split(","))
for c in range(0, max(show_at) + 1):
    ops.step(0, c)
    if c in show_at:
        t = show("X(%d)" % c, ["start", "LDS+sync", "own drain done", "walk done", "flushed", "end"])
        if c >= 1:
            print("   narrow(%d): stats+compaction %.2f | moments staged %.2f | atom 0 done %.2f | all atoms %.2f | stores issued %.2f us"
                  % (c - 1, t[1], t[2], t[3], t[4], t[5]))
    if c >= 1:
        ops.step(1, c)
        if c in show_at:
            show("Y(%d)" % c, ["start", "LDS+sync", "collect done", "apply done", "flushed", "end"])
