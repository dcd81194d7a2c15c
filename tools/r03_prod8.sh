R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_pc; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_restore_gpu.py tests/test_tiled_gpu.py -x -q -m gpu > $O/tests_p8.log 2>&1 || { tail -30 $O/tests_p8.log; exit 1; }
tail -3 $O/tests_p8.log
ABL_VAR=C3_PROD8 ABL_FILE=conv_pc.hip ABL_KERNEL=conv_pc_kernel ABL_VALUES="1 0 1 0" bash tools/s2_buildab.sh
python -m image_restoration_platform_amd.build > /dev/null 2>&1
