# GPU box: kernel-trace stats of the restore step at batch 1, 2, 4, 8 (1024^2): does a level-0 tensor that fits the 256 MiB Infinity
# Cache (67 MB per image) shorten the level-0 kernels per image?   usage: bash tools/r03_bs.sh
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_bs; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for b in 1 2 4 8; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/b$b -o r --output-format csv -- python3 $R/bench.py --batch $b --steps 10 --warmup 2 --no-cpu-baseline --no-host-path --no-profile > $O/b$b.log 2>&1
  echo "batch $b"; tail -1 $O/b$b.log | cut -c1-160
  python3 - $O/b$b $b <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
b = int(sys.argv[2])
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "conv" in n or "classifier" in n:
        short = n.split("::")[-1].split("(")[0]
        print(f"  {short:50s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us  per image {float(r['AverageNs'])/1e3/b:8.1f}")
PY
done
