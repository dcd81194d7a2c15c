# GPU box: the composed up+fuse kernel -- layer/end-to-end/tiled tests, then same-box A/B of IRE_UP_FUSE
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s2; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_restore_gpu.py tests/test_tiled_gpu.py -x -q -m gpu > $O/upf_tests.log 2>&1; tail -15 $O/upf_tests.log
for v in 1 0 1 0; do
  IRE_UP_FUSE=$v timeout -k 10 200 python bench.py --no-cpu-baseline --profile-all > $O/upf_$v.json 2>> $O/upf_err.log || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/s2/upf_$v.json").read().strip().splitlines()[-1])
print("IRE_UP_FUSE=$v", round(d["value"],1), round(d["ms_per_step"],3), d["roofline"]["frac"], {k: round(v,3) for k,v in d["roofline"]["family_ms_per_step"].items()})
PY
done
