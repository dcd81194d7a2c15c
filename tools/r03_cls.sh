R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_cls; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_classifier_gpu.py tests/test_tiled_gpu.py -x -q -m gpu > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout -k 10 200 python bench.py --workload classify --steps 50 --no-cpu-baseline > $O/bench_classify.json 2>$O/err.log; python3 -c "
import json; d=json.loads(open('$O/bench_classify.json').read().strip().splitlines()[-1]); print('classify', round(d['value']), 'img/s', d['roofline']['family_ms_per_step']*1e3, 'us', round(d['roofline']['frac'],4))"
