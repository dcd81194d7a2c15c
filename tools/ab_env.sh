# same-box A/B of run-time engine switches: AB_LIST="name ENV=VAL;name2 ENV2=VAL2" (bench with per-family profile, interleaved twice)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_env; mkdir -p $O
cd $R
for rep in 1 2; do
  echo "$AB_LIST" | tr ';' '\n' | while read -r name envs; do
    [ -z "$name" ] && continue
    env $envs timeout -k 10 200 python bench.py --no-cpu-baseline --profile-all > $O/b_${name}_$rep.json 2>> $O/err.log
    python -c "
import json; j=json.load(open('$O/b_${name}_$rep.json')); print('$name', round(j['value'],1), {k:round(x,3) for k,x in j['roofline']['family_ms_per_step'].items()})"
  done
done
