R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_stamps; mkdir -p $O; cd $R
touch image_restoration_platform_amd/csrc/conv_pc.hip image_restoration_platform_amd/csrc/conv_w4.hip image_restoration_platform_amd/csrc/conv_rb.hip
env IRE_RB_ABLATE=2 python -m image_restoration_platform_amd.build > $O/pcstamps_build.log 2>&1 || { tail $O/pcstamps_build.log; exit 1; }
for v in 64 64r 32; do
  env IRE_RB_STAMPS=$v timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-path --no-profile > $O/pcstamps_$v.json 2> $O/pcstamps_$v.err
  echo "== $v"; grep "stamps\] wg 0\|raw deltas" $O/pcstamps_$v.err | awk '/stamps/{n=0; print; next} {if (n++ < 12) print}' | head -30
done
touch image_restoration_platform_amd/csrc/conv_pc.hip image_restoration_platform_amd/csrc/conv_w4.hip image_restoration_platform_amd/csrc/conv_rb.hip
python -m image_restoration_platform_amd.build > /dev/null 2>&1
