R=$GRAFT_REPO_ROOT; cd $R
ABL_VAR=C3_RES_PRE ABL_FILE=conv_pc.hip ABL_KERNEL=conv_pc_kernel ABL_VALUES="2 1 3 4 0 2" bash tools/s2_buildab.sh
python -m image_restoration_platform_amd.build > /dev/null 2>&1
