# run on the GPU box: tests, default bench, kernel-trace stats, two PMC passes (each its own run)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout -k 10 200 python bench.py --size 512 --no-cpu-baseline > $O/bench_512.json 2>> $O/bench_default.err
timeout -k 10 200 python bench.py --no-profile --no-cpu-baseline > $O/bench_noprofile.json 2>> $O/bench_default.err
timeout -k 10 200 python bench.py --streams 2 --no-cpu-baseline > $O/bench_2lanes.json 2>> $O/bench_default.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o r --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o r --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $O/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o r --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $O/pmc_write.log 2>&1
tail -3 $O/gpu_tests.log; cat $O/bench_default.json | cut -c1-300
