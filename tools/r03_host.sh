# GPU box: the host path (ire_submit / ire_poll batcher, Node seams) -- its tests, then its rates
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_host; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_restore_gpu.py tests/test_node_adapter.py tests/test_serving.py tests/test_torch_ext.py -x -q -m gpu > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout -k 10 300 python tools/host_path_rate.py > $O/rate.log 2>&1; cat $O/rate.log
