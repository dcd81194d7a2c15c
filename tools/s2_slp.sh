# GPU box: same-box A/B of the SLP vectorizer (IRE_SLP=1: hipcc packs f32 pairs into v_pk_*_f32) over the whole library
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s2; mkdir -p $O
for v in 0 1 0 1; do
  cd $R; env IRE_SLP=$v python -m image_restoration_platform_amd.build > /dev/null 2>&1 || { echo build failed; exit 1; }
  if [ "$v" = 0 ] && [ ! -f /tmp/slp_tested ]; then timeout -k 10 900 python -m pytest tests/test_restore_gpu.py tests/test_tiled_gpu.py -x -q -m gpu > $O/slp_tests.log 2>&1; tail -3 $O/slp_tests.log; touch /tmp/slp_tested; fi
  cd /tmp; export TMPDIR=/tmp; rm -rf $O/slp_$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/slp_$v -o r --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/slp_$v.log 2>&1
  cd $R; timeout -k 10 200 python bench.py --no-cpu-baseline --no-profile > $O/slp_$v.json 2>> $O/slp_err.log
  python3 - <<PY
import csv, glob, json
f = glob.glob("$O/slp_$v/**/r_kernel_stats.csv", recursive=True)[0]
d = json.loads(open("$O/slp_$v.json").read().strip().splitlines()[-1])
print("IRE_SLP=$v", round(d["value"], 1), "img/s |", " ".join("%s=%.1f" % (r["Name"].replace("void ire::(anonymous namespace)::","").replace("ire::(anonymous namespace)::","").replace("(ire::ConvArgs)","")[:30], float(r["AverageNs"])/1e3) for r in list(csv.DictReader(open(f)))[:13]))
PY
done
cd $R; python -m image_restoration_platform_amd.build > /dev/null 2>&1
