# GPU box: pre-activated input (separate gn_apply_silu pass) for C >= 256 / C >= 128 vs the fused default: per-kernel stats + img/s
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_split; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in 512 256 128; do
  rm -rf $O/s_$v
  env IRE_ACT_SPLIT_MINC=$v timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/s_$v -o r --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path > $O/s_$v.log 2>&1
  env IRE_ACT_SPLIT_MINC=$v timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline --no-host-path --no-profile > $O/s_$v.json 2>/dev/null
  python3 - <<PY
import csv, glob, json
f = glob.glob("$O/s_$v/**/r_kernel_stats.csv", recursive=True)[0]
d = json.loads(open("$O/s_$v.json").read().strip().splitlines()[-1])
print("ACT_SPLIT_MINC=$v", round(d["value"], 1), "img/s |", " ".join("%s x%s=%.1f" % (r["Name"][r["Name"].find("conv_w4") if "conv_w4" in r["Name"] else r["Name"].find("gn_"):][:40], r["Calls"], float(r["AverageNs"])/1e3) for r in csv.DictReader(open(f)) if "conv_w4" in r["Name"] or "gn_apply" in r["Name"]))
PY
done
