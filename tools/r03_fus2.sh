# GPU box: the fusion chain's kernels by SAD grid cap (IRE_FUSE_GCAP).   usage: bash tools/r03_fus2.sh
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_fus2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cap in 512 64 32 16; do
  export IRE_FUSE_GCAP=$cap
  timeout -k 10 200 rocprofv3 --kernel-trace -d $O/c$cap -o r --output-format csv -- python3 $R/bench.py --workload fusion --steps 10 --no-cpu-baseline > $O/c$cap.log 2>&1
  echo "cap $cap"
  python3 - $O/c$cap <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "fusion" not in n and "Fuse" not in n: continue
    key = (n.split("::")[-1][:28], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: kv[0][3] + kv[0][2]):
    v.sort(); print("  ", k, len(v), "med %.1f" % v[len(v) // 2], "min %.1f" % v[0])
PY
done
