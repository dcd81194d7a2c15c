# GPU box: tests, then the default bench and the secondary workloads (each its own process)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r02_check}
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
cut -c1-400 $O/bench_default.json
timeout -k 10 200 python bench.py --workload fusion --steps 10 --no-cpu-baseline > $O/bench_fusion.json 2>> $O/bench_default.err
cut -c1-300 $O/bench_fusion.json
timeout -k 10 200 python bench.py --workload classify --steps 50 --no-cpu-baseline > $O/bench_classify.json 2>> $O/bench_default.err
cut -c1-300 $O/bench_classify.json
