# GPU box: timing ablations of conv_pc (C3_ABL build macro; results wrong by design): per-kernel rocprof averages
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_pcabl; mkdir -p $O; cd $R
for v in 0 1 2 4 8 16 32 3 7 15; do
  touch image_restoration_platform_amd/csrc/conv_pc.hip
  env C3_ABL=$v python -m image_restoration_platform_amd.build > /dev/null 2>&1 || { echo build failed; exit 1; }
  cd /tmp; export TMPDIR=/tmp; rm -rf $O/abl_$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/abl_$v -o r --output-format csv -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-path --no-profile > $O/abl_$v.log 2>&1
  cd $R
  python3 - <<PY
import csv, glob
f = glob.glob("$O/abl_$v/**/r_kernel_stats.csv", recursive=True)[0]
print("C3_ABL=$v", " ".join("%s=%.1f" % (r["Name"].split("conv_pc_kernel")[1][:18], float(r["AverageNs"])/1e3) for r in csv.DictReader(open(f)) if "conv_pc_kernel" in r["Name"]))
PY
done
touch image_restoration_platform_amd/csrc/conv_pc.hip; python -m image_restoration_platform_amd.build > /dev/null 2>&1
