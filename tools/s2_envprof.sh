# GPU box: per-kernel stats of the default bench under env settings given as arguments: bash tools/s2_envprof.sh "A=1 B=2" "C=3"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s2; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for e in "$@"; do
  i=$((i+1)); rm -rf $O/ep_$i
  env $e timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/ep_$i -o r --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/ep_$i.log 2>&1
  (cd $R; env $e timeout -k 10 200 python bench.py --no-cpu-baseline --no-profile > $O/ep_$i.json 2>> $O/ep_err.log)
  python3 - <<PY
import csv, glob, json
f = glob.glob("$O/ep_$i/**/r_kernel_stats.csv", recursive=True)[0]
d = json.loads(open("$O/ep_$i.json").read().strip().splitlines()[-1])
print("== $e :", round(d["value"], 1), "img/s")
for r in list(csv.DictReader(open(f)))[:14]:
    print("   %-100s %4s %8.1f" % (r["Name"].replace("ire::(anonymous namespace)::","")[:100], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
