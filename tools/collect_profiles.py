"""Copy one round's measurement artifacts from gpurun_out/ (scratch) into profiles/ (tracked) and refresh the two JSON files
bench.py reads (profiles/kernel_durations.json, profiles/traffic.json).  usage: python tools/collect_profiles.py r05
Expects the outputs of tools/profile_all.sh <tag> (+ optionally gpurun_out/<tag>_gpu_tests.log, <tag>_soak_sweep.txt)."""
import json, os, re, shutil, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = lambda *p: os.path.join(root, "gpurun_out", *p)
pr = lambda *p: os.path.join(root, "profiles", *p)
for src, dst in [("prof_%s_summary.txt" % tag, "%s_rocprofv3_bench_summary.txt" % tag),
                 ("prof_aux_%s_summary.txt" % tag, "%s_rocprofv3_aux_summary.txt" % tag),
                 ("prof_ksvd_%s_summary.txt" % tag, "%s_rocprofv3_ksvd_summary.txt" % tag),
                 ("prof_exact_%s_summary.txt" % tag, "%s_rocprofv3_exact_ksvd_summary.txt" % tag),
                 ("bench_%s.json" % tag, "%s_bench_n1.json" % tag),
                 ("%s_gpu_tests.log" % tag, "%s_gpu_tests.log" % tag),
                 ("%s_soak_sweep.txt" % tag, "%s_soak_sweep.txt" % tag)]:
    if os.path.exists(g(src)):
        shutil.copy(g(src), pr(dst))
        print("copied", src, "->", dst)
old = json.load(open(pr("kernel_durations.json")))
new = json.load(open(g("kernel_durations_%s.json" % tag)))
aux = json.load(open(g("kernel_durations_aux_%s.json" % tag)))
for k, v in aux.items():
    if k.startswith("bomp_block_kernel_") or k.startswith("lasso_coder_"):
        new[k] = v
new["aux_source"] = old.get("aux_source")
json.dump(new, open(pr("kernel_durations.json"), "w"), indent=1)
t = json.load(open(pr("traffic.json")))
t["bomp_wave_kernel_bytes_per_launch"] = new["bomp_wave_kernel_bytes_per_launch"]
t["alpha0_n64_kernel_bytes_per_launch"] = new["alpha0_n64_kernel_bytes_per_launch"]
t["tcc"]["bomp_wave_kernel"] = {"hit": new["bomp_wave_kernel_tcc_hit"], "miss": new["bomp_wave_kernel_tcc_miss"]}
t["tcc"]["alpha0_n64_kernel"] = {"hit": new["alpha0_n64_kernel_tcc_hit"], "miss": new["alpha0_n64_kernel_tcc_miss"]}
txt = open(g("prof_ksvd_%s_summary.txt" % tag)).read()


def grab(kern, key):
    return float(re.search(re.escape(kern) + r".*?" + key + r"=([0-9.e+]+)", txt).group(1))


step, fin = "bksvd_step_kernel<1, 3, 1, 64, true>", "bksvd_final_kernel<1, 3, 1, true>"
f, w, ff, fw = grab(step, "FETCH_SIZE"), grab(step, "WRITE_SIZE"), grab(fin, "FETCH_SIZE"), grab(fin, "WRITE_SIZE")
nl = int(re.search(r"last sweep: (\d+) launches", txt).group(1))
b = t["bksvd_step_kernel"]
b.update({"fetch_size_kb_per_launch": f, "write_size_kb_per_launch": w, "bytes_per_launch": (2 * f + w) * 1024,
          "launches_per_sweep": nl, "final_kernel_bytes": (2 * ff + fw) * 1024})
b["bytes_per_sweep"] = b["bytes_per_launch"] * nl + b["final_kernel_bytes"]
b["ratio"] = b["bytes_per_sweep"] / b["algorithmic_bytes_per_sweep_survey_8d"]
b["final_kernel_note"] = "bksvd_final_kernel by its own counters (FETCH_SIZE %.1f MB, WRITE_SIZE %.1f MB per launch)" % (ff / 1e3, fw / 1e3)
al = t.get("aux_legs", {})
if "bomp_block_kernel" in al:
    al["bomp_block_kernel"]["bytes_per_launch"] = new["bomp_block_kernel_bytes_per_launch"]
if "lasso_coder" in al:
    al["lasso_coder"]["bytes_per_launch"] = new["lasso_coder_bytes_per_launch"]
json.dump(t, open(pr("traffic.json"), "w"), indent=1)
print("sweep: %d launches, %.3f GB per sweep, ratio %.3f" % (nl, b["bytes_per_sweep"] / 1e9, b["ratio"]))
