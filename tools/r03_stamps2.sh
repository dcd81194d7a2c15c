R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_stamps; mkdir -p $O; cd $R
for v in 128 128r 256r; do echo "== stamps $v"; bash tools/s2_stamps.sh $v > $O/stamps2_$v.log 2>&1; grep -B1 -A1 "epilogue:" $O/stamps2_$v.log | head -24; done
python -m image_restoration_platform_amd.build > /dev/null 2>&1
