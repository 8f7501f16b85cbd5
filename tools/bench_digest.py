"""The numbers of a bench.py JSON line that DESIGN.md quotes.  usage: python tools/bench_digest.py <bench.json>"""
import json
import sys

d = json.load(open(sys.argv[1]))
r, k = d["roofline"], d.get("ksvd_iteration", {})
print("value %.1f M patches/s | %.3f ms per step | greedy %.3f ms frac %.4f | gemm %.3f ms | whole step %.4f | sclk %s MHz | traffic_stale %s"
      % (d["value"] / 1e6, d["ms_per_step"], r["avg_launch_ms"], r["frac"], r["gemm_stage"]["avg_launch_ms"], r["whole_step"]["frac"],
         ("%.0f" % r["sclk_mhz"]["step_loop"]) if r.get("sclk_mhz", {}).get("step_loop") else "-", r.get("traffic_stale")))
if k:
    ms = k["ms"]
    print("alternation: encode %.2f residual %.2f sweep %.2f error %.2f = %.2f ms | fifty %.1f ms | sweep frac %.4f | exact sweep %s ms | sclk beside "
          "the alternation's encode %s MHz" % (ms["encode"], ms["residual"], ms["sweep"], ms["error"], k["ms_total"],
                                             k.get("fifty_iterations", {}).get("ms_total", float("nan")), k["sweep_roofline"]["frac"],
                                             ("%.2f" % k["exact_sweep"]["ms"]) if "exact_sweep" in k and "ms" in k["exact_sweep"] else "-",
                                             ("%.0f" % k["sclk_mhz_encode"]) if k.get("sclk_mhz_encode") else "-"))
if "config3_shard" in d:
    print("config3_shard %.2f M patches/s | config4 LARS %.1f ms" % (d["config3_shard"]["value"] / 1e6, d["config4_minibatch"]["ms"]["lars_coder"]))
if "cpu_baseline" in d:
    c = d["cpu_baseline"]
    print("cpu: %.0f patches/s on %d core | map %.0f | C/OpenMP %.0f" % (c["value"], c["cores"], c.get("all_cores_value", 0), c.get("c_port_value", 0)))
