# GPU box: kernel-trace stats of the default bench (per-kernel average durations)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s2; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rm -rf $O/stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o r --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline ${BENCH_ARGS} > $O/stats.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$O/stats/**/r_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:24]:
    print("%-110s %5s %9.1f %6.2f" % (r["Name"][:110], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
