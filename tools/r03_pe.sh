# GPU box: the default bench with the family's launches timed on every step / every 4th / not at all (same box, alternating).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_pe; mkdir -p $O; cd $R
for v in 1 4 0 1 4 0; do
  if [ $v = 0 ]; then x="--no-profile"; else x="--profile-every $v"; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-host-path $x > $O/b_$v.json 2>> $O/err.log
  python3 -c "
import json; d=json.loads(open('$O/b_$v.json').read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('every $v:', round(d['value'],1), 'img/s frac', round(r.get('frac',0),4), 'launches', r.get('launches'), 'avg us', round(r.get('avg_launch_us',0),1), 'L0 ms', round(r.get('per_level',{}).get('L0',{}).get('ms_per_step',0),3))"
done
