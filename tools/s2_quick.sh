# GPU box: quick check (restore + tiled tests) and two default benches with per-kernel trace:  bash tools/s2_quick.sh [pytest -k expr]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s2; mkdir -p $O; cd $R
timeout -k 10 800 python -m pytest tests/test_restore_gpu.py tests/test_tiled_gpu.py -x -q -m gpu ${1:+-k "$1"} > $O/q_tests.log 2>&1; tail -5 $O/q_tests.log
for i in 1 2; do
timeout -k 10 200 python bench.py --no-cpu-baseline --profile-all > $O/q_$i.json 2>> $O/q_err.log || exit 1
python - <<PY
import json
d=json.loads(open("gpurun_out/s2/q_$i.json").read().strip().splitlines()[-1])
print(round(d["value"],1), round(d["ms_per_step"],3), round(d["roofline"]["frac"],4), {k: round(v,3) for k,v in d["roofline"]["family_ms_per_step"].items()})
PY
done
bash tools/s2_prof.sh
