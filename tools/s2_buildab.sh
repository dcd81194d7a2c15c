# GPU box: same-box A/B of a BUILD-time macro:  ABL_VAR=W4_STAGGER ABL_FILE=conv_w4.hip ABL_KERNEL=conv_w4_kernel ABL_VALUES="1 0 1 0" bash tools/s2_buildab.sh
# per value: rebuild that file, rocprofv3 kernel stats of the default bench (avg us of the matching kernels) + img/s
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s2; mkdir -p $O
for v in ${ABL_VALUES}; do
  cd $R; touch image_restoration_platform_amd/csrc/$ABL_FILE
  env $ABL_VAR=$v python -m image_restoration_platform_amd.build > /dev/null 2>&1 || { echo build failed; exit 1; }
  if [ -n "$ABL_TESTS" ]; then timeout -k 10 600 python -m pytest $ABL_TESTS -x -q -m gpu > $O/bab_tests_$v.log 2>&1; tail -3 $O/bab_tests_$v.log; fi
  cd /tmp; export TMPDIR=/tmp; rm -rf $O/bab_$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/bab_$v -o r --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path > $O/bab_$v.log 2>&1
  cd $R; timeout -k 10 200 python bench.py --no-cpu-baseline --no-host-path --no-profile > $O/bab_$v.json 2>> $O/bab_err.log
  python3 - <<PY
import csv, glob, json
f = glob.glob("$O/bab_$v/**/r_kernel_stats.csv", recursive=True)[0]
d = json.loads(open("$O/bab_$v.json").read().strip().splitlines()[-1])
print("$ABL_VAR=$v", round(d["value"], 1), "img/s |", " ".join("%s=%.1f" % (r["Name"].split("$ABL_KERNEL")[1][:34], float(r["AverageNs"])/1e3) for r in csv.DictReader(open(f)) if "$ABL_KERNEL" in r["Name"]))
PY
done
cd $R; touch image_restoration_platform_amd/csrc/$ABL_FILE
