"""From a rocprofv3 kernel-trace (rocpd .db) of tools/ksvd_bench.py: durations of the block-sweep launches of the last
sweep.  usage: step_durations.py <trace dir> <launches per sweep>: 129 (default) = X(0), then one MERGED launch per block
[narrow(c-1) || X(c)] -> flag -> [Y(c)], c = 1 .. 128 at K = 1024, B = 8 (round 5); 257 = LYS_BKSVD_MERGED=0: X(0), then X(c), Y(c)."""
import glob, sqlite3, sys
import numpy as np
f = sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True))[0]
db = sqlite3.connect(f)
rows = db.execute("select name, start, end from kernels order by start").fetchall()
step = [(s, e) for (nm, s, e) in rows if "bksvd_step_kernel" in nm]
per = int(sys.argv[2]) if len(sys.argv) > 2 else 129
last = step[-per:]
d = np.array([e - s for s, e in last]) / 1e3
gaps = np.array([last[i + 1][0] - last[i][1] for i in range(len(last) - 1)]) / 1e3
print("last sweep: %d launches, total kernel %.1f us, span %.1f us, gaps mean %.2f us (sum %.1f)" %
      (len(d), d.sum(), (last[-1][1] - last[0][0]) / 1e3, gaps.mean(), gaps.sum()))
if per % 2 == 1 and per > 200:   # two launches per block
    X = np.concatenate([[d[0]], d[1::2]])   # X(0), X(1), Y(1), X(2), Y(2), ...
    Y = d[2::2]
    print("X: mean %.1f  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f us" % (X.mean(), *np.percentile(X, [10, 50, 90]), X.max()))
    print("Y" + ": mean %.1f  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f us" % (Y.mean(), *np.percentile(Y, [10, 50, 90]), Y.max()))
    print("X first 12:", np.round(X[:12], 1).tolist())
    print("Y first 12:", np.round(Y[:12], 1).tolist())
else:                            # X(0), then the merged launches, the last one = the narrow step of the last block alone
    M = d[1:-1]
    print("X(0) %.1f us | merged launches: mean %.1f  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f us | last (narrow step only) %.1f us"
          % (d[0], M.mean(), *np.percentile(M, [10, 50, 90]), M.max(), d[-1]))
    print("first 12 merged:", np.round(M[:12], 1).tolist())
