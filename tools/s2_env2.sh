# GPU box: process-level env A/B of the default bench (runtime knobs): bash tools/s2_env2.sh "VAR=val" ...   ("-" = none)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s2; mkdir -p $O; cd $R
for rep in 1 2; do
for e in "$@"; do
  if [ "$e" = "-" ]; then v=""; else v="$e"; fi
  env $v timeout -k 10 200 python bench.py --no-cpu-baseline --no-profile > $O/env2.json 2>> $O/env2_err.log || exit 1
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/s2/env2.json").read().strip().splitlines()[-1])
print("[$e]", round(d["value"],1), "img/s", round(d["ms_per_step"],3), "ms")
PY
done; done
