# GPU box: kernel-trace stats of the fusion workload (cfg 3 + its batched entry).   usage: bash tools/r03_fus.sh
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_fus; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/st -o r --output-format csv -- python3 $R/bench.py --workload fusion --steps 10 --no-cpu-baseline > $O/st.log 2>&1
tail -1 $O/st.log | cut -c1-200
python3 - $O/st <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "fusion" in n:
        print(f"  {n[:90]:90s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
