# GPU box: classifier tests (bit-exact vs the oracle) + the classify bench + everything that classifies inside
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s2; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_classifier_gpu.py tests/test_abi.py tests/test_restore_gpu.py -x -q -m gpu > $O/cls_tests.log 2>&1; tail -8 $O/cls_tests.log
timeout -k 10 200 python bench.py --workload classify --steps 50 --no-cpu-baseline > $O/cls_bench.json 2>> $O/cls_err.log; cut -c1-900 $O/cls_bench.json
timeout -k 10 200 python bench.py --no-cpu-baseline --profile-all > $O/cls_def.json 2>> $O/cls_err.log
python - <<PY
import json
d=json.loads(open("gpurun_out/s2/cls_def.json").read().strip().splitlines()[-1])
print(round(d["value"],1), round(d["ms_per_step"],3), round(d["roofline"]["frac"],4), {k: round(v,3) for k,v in d["roofline"]["family_ms_per_step"].items()})
PY
