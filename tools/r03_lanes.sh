# GPU box: 1 lane vs 2 lanes (two HIP streams, half a batch each), unprofiled, alternating.   usage: bash tools/r03_lanes.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_lanes; mkdir -p $O; cd $R
for v in 1 2 1 2; do
  timeout -k 10 200 python bench.py --streams $v --no-cpu-baseline --no-host-path --no-profile > $O/l_$v.json 2>> $O/err.log
  python3 -c "
import json; d=json.loads(open('$O/l_$v.json').read().strip().splitlines()[-1]); print('lanes $v:', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms')"
done
