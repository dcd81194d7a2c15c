# GPU box: conv_upq (IRE_UPQ bit 0: level 2, bit 1: level 1) against conv_up: restore + tiled parity suites, then same-box bench A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_upq; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_restore_gpu.py tests/test_tiled_gpu.py -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
for rep in 1 2; do for q in ${UPQ_LIST:-3 1 0}; do
  IRE_UPQ=$q timeout -k 10 200 python bench.py --no-cpu-baseline --no-host-path --no-secondary > $O/bench_$q.json 2> $O/bench_$q.err || { tail -5 $O/bench_$q.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/bench_$q.json").read().strip().splitlines()[-1])
print("IRE_UPQ=$q", round(d["value"],1), "img/s;", {g["group"]: round(g["us_per_launch"], 1) for g in d["roofline"].get("per_group", []) if g["group"][:2] in ("up", "fu")})
PY
done; done
