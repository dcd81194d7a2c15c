# GPU box: timing ablations of conv_up.hip's fused epilogue (IRE_UP_ABL bits; results of ablated builds are wrong by design)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s2; mkdir -p $O
for v in ${ABL_VALUES:-0 1 2 4 7}; do
  cd $R; touch image_restoration_platform_amd/csrc/conv_up.hip
  env ${ABL_VAR:-IRE_UP_ABL}=$v python -m image_restoration_platform_amd.build > /dev/null 2>&1 || { echo build failed; exit 1; }
  cd /tmp; export TMPDIR=/tmp; rm -rf $O/abl_$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/abl_$v -o r --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/abl_$v.log 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob("$O/abl_$v/**/r_kernel_stats.csv", recursive=True)[0]
print("${ABL_VAR:-IRE_UP_ABL}=$v", " ".join("%s=%.1f" % (r["Name"].split("conv_up_kernel")[1][:3], float(r["AverageNs"])/1e3) for r in csv.DictReader(open(f)) if "conv_up_kernel" in r["Name"]))
PY
done
cd $R; touch image_restoration_platform_amd/csrc/conv_up.hip
