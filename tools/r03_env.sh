# GPU box: same-box A/B of engine env settings, unprofiled default bench, alternating:  bash tools/r03_env.sh "A=1" "A=0" ...
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_env; mkdir -p $O; cd $R
i=0
for rep in 1 2; do
for e in "$@"; do
  i=$((i+1))
  env $e timeout -k 10 200 python bench.py --no-cpu-baseline --no-host-path --no-profile > $O/e_$i.json 2>> $O/err.log
  python3 -c "
import json; d=json.loads(open('$O/e_$i.json').read().strip().splitlines()[-1]); print('$e :', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms')"
done
done
