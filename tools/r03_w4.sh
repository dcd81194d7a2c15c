# GPU box: conv_w4 line-coalesced epilogue: tests, then same-box build A/B (IRE_W4_TEPI = 1 / 0)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_w4; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_restore_gpu.py tests/test_tiled_gpu.py -x -q -m gpu > $O/tests.log 2>&1; tail -4 $O/tests.log
ABL_VAR=IRE_W4_TEPI ABL_FILE=conv_w4.hip ABL_KERNEL=conv_w4_kernel ABL_VALUES="1 0 1 0" bash tools/s2_buildab.sh
python -m image_restoration_platform_amd.build > /dev/null 2>&1
