# GPU box: tests, then same-box interleaved A/B of one engine switch:  bash tools/s2_ab.sh VAR [tests]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s2; mkdir -p $O; cd $R
VAR=$1; T=${2:-tests/test_restore_gpu.py tests/test_tiled_gpu.py}
if [ "$T" != "none" ]; then
timeout -k 10 800 python -m pytest $T -x -q -m gpu > $O/ab_tests.log 2>&1; tail -8 $O/ab_tests.log
fi
for v in 1 0 1 0; do
  env $VAR=$v timeout -k 10 200 python bench.py --no-cpu-baseline --profile-all > $O/ab_$v.json 2>> $O/ab_err.log || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/s2/ab_$v.json").read().strip().splitlines()[-1])
print("$VAR=$v", round(d["value"],1), round(d["ms_per_step"],3), round(d["roofline"]["frac"],4), {k: round(v,3) for k,v in d["roofline"]["family_ms_per_step"].items()})
PY
done
