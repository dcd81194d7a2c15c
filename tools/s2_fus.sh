# GPU box: per-kernel stats of the fusion workload (cfg 3) + its bench line + the fusion parity tests.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s2; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/fus_build.log 2>&1 || { tail -20 $O/fus_build.log; exit 1; }
timeout -k 10 400 python -m pytest tests/test_fusion_gpu.py -x -q -m gpu > $O/fus_tests.log 2>&1; tail -3 $O/fus_tests.log
cd /tmp; export TMPDIR=/tmp; rm -rf $O/fus_prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/fus_prof -o r --output-format csv -- python3 $R/bench.py --workload fusion --steps 10 --warmup 2 --no-cpu-baseline > $O/fus_prof.log 2>&1
cd $R; timeout -k 10 200 python bench.py --workload fusion --no-cpu-baseline > $O/fus.json 2>> $O/fus_err.log
python3 - <<PY
import csv, glob, json
f = glob.glob("$O/fus_prof/**/r_kernel_stats.csv", recursive=True)[0]
d = json.loads(open("$O/fus.json").read().strip().splitlines()[-1])
print("fusion:", round(d["value"], 1), d["unit"], "ms/step", d["ms_per_step"], "roofline", d.get("roofline"))
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "fusion" in n or "cls" in n or "classif" in n:
        print("   %-90s %4s %8.1f" % (n.replace("ire::(anonymous namespace)::","")[:90], r["Calls"], float(r["AverageNs"])/1e3))
PY
