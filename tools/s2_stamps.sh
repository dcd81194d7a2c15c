# GPU box: s_memtime stamps of conv_w4's stage phases (diagnostic build: IRE_RB_ABLATE=2; results are printed at engine teardown).
# bash tools/s2_stamps.sh "<IRE_RB_STAMPS value, e.g. 128 or 128r>"   (prints both stamped waves of workgroup 0: the older and the younger wave of a SIMD)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s2; mkdir -p $O; cd $R
touch image_restoration_platform_amd/csrc/conv_w4.hip image_restoration_platform_amd/csrc/conv_rb.hip
env IRE_RB_ABLATE=2 python -m image_restoration_platform_amd.build > $O/stamps_build.log 2>&1 || { tail $O/stamps_build.log; exit 1; }
env IRE_RB_STAMPS=$1 timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $O/stamps.json 2> $O/stamps.err
awk '/^\[stamps\] wg 0 wave 0/{f=1} /^\[stamps\] wg 1/{f=0} f' $O/stamps.err | grep "stamps\|k-steps\|stage=" | awk '/stamps/{n=0} {if (n++ < 21) print}' 
touch image_restoration_platform_amd/csrc/conv_w4.hip image_restoration_platform_amd/csrc/conv_rb.hip
