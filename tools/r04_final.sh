# GPU box: the whole -m gpu suite, then round 4's profile set (tools/r04_profile.sh)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_final; mkdir -p $O; cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -4 $O/pytest.log; [ $rc -eq 0 ] || exit $rc
bash tools/r04_profile.sh ${1:-v1}
