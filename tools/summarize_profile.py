"""Condense rocprofv3 outputs (rocpd sqlite .db, the default format of this rocprofv3) into a text summary."""
import glob
import os
import sqlite3
import sys

root = sys.argv[1]
json_out = sys.argv[2] if len(sys.argv) > 2 else None   # e.g. profiles/kernel_durations.json (read by bench.py)
durations = {}
pmc = {}   # kernel field -> counter -> per-dispatch average


def short(name):
    name = name.split("(")[0]
    return name[-84:]


for f in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
    db = sqlite3.connect(f)
    cur = db.cursor()
    tag = os.path.basename(os.path.dirname(f))
    has_pmc = cur.execute("select count(*) from pmc_events").fetchone()[0] > 0
    if not has_pmc:
        print("== rocprofv3 --kernel-trace --stats :: %s ==" % os.path.relpath(f, root))
        print("%-86s %7s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        for name, calls, tot, avg, pct in cur.execute(
                "select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc limit %d" % int(os.environ.get("LYS_SUMMARY_TOP", "16"))):
            print("%-86s %7d %14d %12.0f %7.2f" % (short(name), calls, tot, avg, pct))
            for key, field in (("bomp_wave2_kernel", "bomp_wave_kernel"), ("bomp_wave_kernel", "bomp_wave_kernel"),
                               ("alpha0_n64_bf16x3_kernel", "alpha0_n64_kernel"),
                               ("alpha0_n64_kernel", "alpha0_n64_kernel"), ("bksvd_step_kernel", "bksvd_step_kernel")):
                if key in name and field + "_avg_ms" not in durations:
                    durations[field + "_avg_ms"] = avg / 1e3   # top_kernels.average is in us
                    durations[field + "_calls"] = calls
                    durations[field + "_name"] = key
        print("\nregister / LDS use per kernel (from the dispatch records):")
        for name, v, a, s, lds, wg, gx in cur.execute(
                "select name,vgpr_count,accum_vgpr_count,sgpr_count,lds_size,workgroup_x,max(grid_x) from kernels group by name order by sum(duration) desc limit 10"):
            print("%-86s vgpr=%s agpr=%s sgpr=%s lds=%s wg=%s max_grid=%s" % (short(name), v, a, s, lds, wg, gx))
    else:
        print("\n== PMC pass %s: per-dispatch averages (sum over instances) ==" % tag)
        q = ("select k.name, e.counter_name, sum(e.counter_value), count(distinct e.dispatch_id), avg(k.duration) "
             "from pmc_events e join kernels k on k.dispatch_id = e.dispatch_id group by k.name, e.counter_name")
        rows = {}
        for name, cname, tot, nd, dur in cur.execute(q):
            rows.setdefault(name, {"n": nd, "dur": dur})[cname] = tot / max(1, nd)
        for name, r in rows.items():
            for key, field in (("bomp_wave2_kernel", "bomp_wave_kernel"), ("bomp_wave_kernel", "bomp_wave_kernel"),
                               ("alpha0_n64", "alpha0_n64_kernel"), ("bksvd_step_kernel", "bksvd_step_kernel"),
                               ("bomp_block_kernel", "bomp_block_kernel"), ("lasso_lars_kernel", "lasso_lars_kernel"),
                               ("lasso_ws_kernel", "lasso_coder")):
                if key in name:
                    pmc.setdefault(field, {}).update({c: v for c, v in r.items() if c not in ("n", "dur")})
                    break
        for name in sorted(rows, key=lambda x: -rows[x]["n"] * rows[x]["dur"])[:int(os.environ.get("LYS_SUMMARY_TOP", "16")) // 2]:
            r = rows[name]
            vals = ", ".join("%s=%.5g" % (c, v) for c, v in sorted(r.items()) if c not in ("n", "dur"))
            print("%-60s dispatches=%d avg_ns=%.0f  %s" % (short(name)[-60:], r["n"], r["dur"], vals))

if json_out:
    import json
    # HBM bytes per launch from the separate FETCH_SIZE / WRITE_SIZE passes (KB units; FETCH_SIZE doubled: it under-counts wide
    # coalesced reads by 2x on gfx950, MI355X_MICROARCH.md HBM section; WRITE_SIZE uncalibrated)
    for field, c in pmc.items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            durations[field + "_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
            durations[field + "_fetch_size_kb"] = c["FETCH_SIZE"]
            durations[field + "_write_size_kb"] = c["WRITE_SIZE"]
        if "TCC_HIT_sum" in c:
            durations[field + "_tcc_hit"] = c["TCC_HIT_sum"]
            durations[field + "_tcc_miss"] = c.get("TCC_MISS_sum")
    durations["source"] = "rocprofv3 --kernel-trace --stats of `python bench.py` (tools/profile.sh), averages over all launches"
    # fingerprint of the kernel sources these figures were taken on: bench.py flags them `traffic_stale` when the tree differs
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_source_sha
    durations["kernel_source_sha"] = kernel_source_sha()
    json.dump(durations, open(json_out, "w"), indent=1)
