"""Condense rocprofv3 outputs (kernel stats + PMC csv) into a small text summary for profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    name = name.split("(")[0]
    return name[-90:]


print("== rocprofv3 --kernel-trace --stats (bench.py --steps 3 --warmup 1) ==")
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    print("file:", os.path.relpath(f, root))
    print("%-92s %8s %14s %12s %8s" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
    for r in rows[:14]:
        print("%-92s %8s %14s %12s %8s" % (short(r.get("Name", "")), r.get("Calls"), r.get("TotalDurationNs"),
                                            r.get("AverageNs"), r.get("Percentage")))
for tag in ("pmc_sq", "pmc_fetch", "pmc_write", "pmc_mfma"):
    files = glob.glob(os.path.join(root, tag, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        agg = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = short(r.get("Kernel_Name", ""))
            agg[k][r.get("Counter_Name")] += float(r.get("Counter_Value", 0) or 0)
            cnt[k].add(r.get("Dispatch_Id"))
        print("\n== PMC pass %s: per-dispatch averages ==" % tag)
        for k in sorted(agg, key=lambda x: -len(cnt[x])):
            n = max(1, len(cnt[k]))
            vals = ", ".join("%s=%.4g" % (c, v / n) for c, v in sorted(agg[k].items()))
            print("%-70s dispatches=%d  %s" % (k[-70:], n, vals))
