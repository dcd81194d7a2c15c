# GPU box: FETCH_SIZE per kernel (MiB per launch, corrected x2) for build-time ablation values of conv_pc.hip
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s2; mkdir -p $O
for v in ${ABL_VALUES:-0}; do
  cd $R; touch image_restoration_platform_amd/csrc/conv_pc.hip
  C3_ABL=$v python -m image_restoration_platform_amd.build > /dev/null 2>&1 || { echo build failed; exit 1; }
  cd /tmp; export TMPDIR=/tmp; rm -rf $O/fe_$v
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/fe_$v -o r --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $O/fe_$v.log 2>&1
  python3 - <<PY
import csv, collections
per = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open("$O/fe_$v/r_counter_collection.csv")):
    if r["Counter_Name"] != "FETCH_SIZE": continue
    per[r["Kernel_Name"]][0] += float(r["Counter_Value"]); per[r["Kernel_Name"]][1] += 1
print("C3_ABL=$v", " | ".join("%s %.0f MiB" % (k.split("conv_pc_kernel")[1][:18], 2 * v[0] / v[1] / 1024) for k, v in per.items() if "conv_pc_kernel" in k))
PY
done
cd $R; touch image_restoration_platform_amd/csrc/conv_pc.hip
