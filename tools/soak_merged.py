"""Race hunt for the merged-launch block sweep: the same sweep (same residual, codes, dictionary) through the merged schedule
(default) and through the two-launch schedule (LYS_BKSVD_MERGED=0), repeated; the two differ only in launch structure, so their
atoms / codes must agree to fp32 summation-order noise.  usage: python tools/soak_merged.py [repeats]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lyssandra_amd import engine as eng

SHAPES = [(64, 1024, 10, 1 << 18), (16, 16, 4, 200000), (8, 8, 3, 100000), (128, 128, 12, 100000), (64, 100, 5, 300000),
          (64, 1030, 10, 100000), (256, 40, 6, 150000), (100, 24, 12, 120000), (32, 64, 8, 400000), (64, 2048, 10, 1 << 19),
          # k > 16 runs the eager schedule whatever the switch says: a run-to-run determinism check of that path
          (128, 128, 20, 100000), (64, 1024, 32, 50000), (200, 64, 8, 100000)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
worst, bad = 0.0, 0
for (n, K, k, N) in SHAPES:
    gen = torch.Generator(device="cuda").manual_seed(11)
    Dt = torch.randn((n, K), device="cuda", generator=gen)
    Dt = Dt / Dt.norm(dim=0, keepdim=True)
    Xs = torch.randn((N, n), device="cuda", generator=gen)
    dd = eng.DeviceDictionary(n, K)
    dd.set(Dt)
    idx, coef0, nnz = eng.bomp_encode(Xs, dd, k)
    coef0 = coef0.clone()
    R0, _ = eng.residual(Xs, dd, idx, coef0, nnz)
    R0 = R0.clone()
    D0 = dd.D.clone()
    ref = None
    wshape = 0.0
    for r in range(reps):
        for merged in ("0", "1"):
            if merged == "0" and r > 0:
                continue   # one reference run of the two-launch schedule per shape
            os.environ["LYS_BKSVD_MERGED"] = merged
            dd.D.copy_(D0)
            dd.invalidate()
            R, coef = R0.clone(), coef0.clone()
            eng.ksvd_cycle(R, dd, idx, coef, nnz, buffers={})
            out = (dd.D.clone(), coef.clone(), R.clone())
            if merged == "0":
                ref = out
            else:
                dD = (out[0] - ref[0]).abs().max().item()
                dc = (out[1] - ref[1]).abs().max().item() / max(ref[1].abs().max().item(), 1e-30)
                dR = (out[2] - ref[2]).abs().max().item() / max(ref[2].abs().max().item(), 1e-30)
                w = max(dD, dc, dR)
                wshape = max(wshape, w)
                if not (w < 2e-5):
                    bad += 1
                    print("MISMATCH n=%d K=%d k=%d N=%d rep %d: atoms %.3g codes %.3g rows %.3g" % (n, K, k, N, r, dD, dc, dR), flush=True)
    worst = max(worst, wshape)
    print("n=%d K=%d k=%d N=%d: %d merged runs, worst difference to the two-launch schedule %.3g" % (n, K, k, N, reps, wshape), flush=True)
print("mismatches: %d, worst %.3g" % (bad, worst))
