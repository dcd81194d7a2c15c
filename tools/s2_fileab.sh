# GPU box: same-box A/B of two versions of one source file:  bash tools/s2_fileab.sh <file in csrc> <alternative copy> <kernel name substring> [pytest args]
# order: new, alt, new, alt; per run: rebuild, (tests for the new one, once), rocprofv3 kernel stats of the default bench + img/s
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s2; mkdir -p $O; F=$R/image_restoration_platform_amd/csrc/$1
cp $F /tmp/fileab_new; cp $R/$2 /tmp/fileab_alt
for v in new alt new alt; do
  cd $R; cp /tmp/fileab_$v $F
  python -m image_restoration_platform_amd.build > /dev/null 2>&1 || { echo build failed; exit 1; }
  if [ "$v" = new ] && [ -n "$4" ] && [ ! -f /tmp/fileab_tested ]; then timeout -k 10 900 python -m pytest $4 -x -q -m gpu > $O/fab_tests.log 2>&1; tail -3 $O/fab_tests.log; touch /tmp/fileab_tested; fi
  cd /tmp; export TMPDIR=/tmp; rm -rf $O/fab_$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/fab_$v -o r --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path > $O/fab_$v.log 2>&1
  cd $R; timeout -k 10 200 python bench.py --no-cpu-baseline --no-host-path --no-profile > $O/fab_$v.json 2>> $O/fab_err.log
  python3 - <<PY
import csv, glob, json
f = glob.glob("$O/fab_$v/**/r_kernel_stats.csv", recursive=True)[0]
d = json.loads(open("$O/fab_$v.json").read().strip().splitlines()[-1])
print("$v", round(d["value"], 1), "img/s |", " ".join("%s=%.1f" % (r["Name"].split("$3")[1][:34], float(r["AverageNs"])/1e3) for r in csv.DictReader(open(f)) if "$3" in r["Name"]))
PY
done
cp /tmp/fileab_new $F
