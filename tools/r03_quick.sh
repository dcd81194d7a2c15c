# GPU box: restore + tiled tests, then kernel stats + img/s of the current build
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_quick; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_restore_gpu.py tests/test_tiled_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
cd /tmp; export TMPDIR=/tmp; rm -rf $O/st
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/st -o r --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path > $O/st.log 2>&1
cd $R; timeout -k 10 200 python bench.py --no-cpu-baseline --no-host-path --no-profile > $O/b.json 2>/dev/null
python3 - <<PY
import csv, glob, json
f = glob.glob("$O/st/**/r_kernel_stats.csv", recursive=True)[0]
d = json.loads(open("$O/b.json").read().strip().splitlines()[-1])
print(round(d["value"], 1), "img/s")
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "conv_" in n or "classifier" in n: print("  %-46s %7.1f" % (n[n.find("conv_") if "conv_" in n else n.find("classifier"):][:46], float(r["AverageNs"])/1e3))
PY
