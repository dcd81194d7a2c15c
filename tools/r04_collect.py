"""Copy the judged summaries of a tools/r04_profile.sh run (gpurun_out/r04_<tag>/) into profiles/:
  python tools/r04_collect.py <tag>
writes r04_<tag>_summary.md, r04_<tag>_kernel_stats_bench_1024_bs8.csv, r04_<tag>_bench_default.json, r04_traffic.json, r04_pmc_util.md,
r04_secondary.json and r04_secondary_<workload>_summary.md (rocprofv3 kernel-trace rows of the secondary workloads)."""
import json
import shutil
import subprocess
import sys

tag = sys.argv[1]
src, dst = "gpurun_out/r04_%s" % tag, "profiles"
last = lambda p: json.loads(open(p).read().strip().splitlines()[-1])
bd = last(src + "/bench_default.json")
shutil.copy(src + "/stats/r_kernel_stats.csv", "%s/r04_%s_kernel_stats_bench_1024_bs8.csv" % (dst, tag))
json.dump(bd, open("%s/r04_%s_bench_default.json" % (dst, tag), "w"), indent=1)
foot = ("Default bench on the same box: %.1f img/s, %.3f ms/step, roofline.frac %.4f, per-layer %.4f (boxes of the pool differ by +-3 %%)."
        % (bd["value"], bd["ms_per_step"], bd["roofline"]["frac"], bd["roofline"].get("frac_of_per_layer_roofline", 0.0)))
run = lambda *a: subprocess.check_call([sys.executable, "tools/profile_summary.py"] + list(a))
run("stats", src + "/stats/r_kernel_stats.csv", "%s/r04_%s_summary.md" % (dst, tag),
    "Round 4 (%s) — rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path --no-secondary` (1024x1024 bs 8, 7 steps traced)" % tag, foot)
run("traffic", src + "/pmc_fetch/r_counter_collection.csv", src + "/pmc_write/r_counter_collection.csv", dst + "/r04_traffic.json", "2")
subprocess.check_call([sys.executable, "tools/pmc_util_summary.py", src + "/pmc", dst + "/r04_pmc_util.md", "Round 4"])
sec = {"512": last(src + "/bench_512.json"), "tiled_bf16": last(src + "/bench_tiled_bf16.json"), "tiled_fp8": last(src + "/bench_tiled_fp8.json"),
       "fusion": last(src + "/bench_fusion.json"), "classify": last(src + "/bench_classify.json"), "noprofile_1024": last(src + "/bench_noprofile.json")}
json.dump(sec, open(dst + "/r04_secondary.json", "w"), indent=1)
for d, name in (("stats_512", "cfg1_512"), ("stats_classifysteps30", "classify"), ("stats_fusionsteps10", "fusion"),
                ("stats_tiledsteps6", "tiled_bf16"), ("stats_tiledprecisionfp", "tiled_fp8")):
    run("stats", "%s/%s/r_kernel_stats.csv" % (src, d), "%s/r04_secondary_%s_summary.md" % (dst, name),
        "Round 4 — rocprofv3 --kernel-trace --stats, secondary workload `%s` (bench.py --workload / --size; same box as r04_%s_summary.md)" % (name, tag))
print("collected", tag)
