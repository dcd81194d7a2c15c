# GPU box: the restore + tiled + fusion GPU tests, then the default bench without its host-side extras (twice).   usage: bash tools/r03_quick.sh
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_quick; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_restore_gpu.py tests/test_tiled_gpu.py tests/test_fusion_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -20 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for i in 1 2; do timeout -k 10 200 python bench.py --no-cpu-baseline --no-host-path > $O/bench_$i.json 2>> $O/err.log; python3 -c "
import json; d=json.loads(open('$O/bench_$i.json').read().strip().splitlines()[-1]); r=d['roofline']; print(round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms frac', round(r['frac'],4), 'exec', round(r.get('frac_executed',0),4))"; done
