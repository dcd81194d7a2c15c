# GPU box: fp8 tests and the fp8 bench in both forms (block-scaled K=64 MFMA / same-rate 32x32x16 fp8)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s2; mkdir -p $O; cd $R
timeout -k 10 300 python -m pytest tests/test_tiled_gpu.py -x -q -m gpu -k fp8 > $O/f8_tests2.log 2>&1; tail -5 $O/f8_tests2.log
IRE_FP8_MX=1 timeout -k 10 200 python bench.py --precision fp8 --no-cpu-baseline --profile-all > $O/b2_mx1.json 2>> $O/err1.log &&
IRE_FP8_MX=0 timeout -k 10 200 python bench.py --precision fp8 --no-cpu-baseline --profile-all > $O/b2_mx0.json 2>> $O/err0.log
python - <<PY
import json
for f in ["b2_mx1","b2_mx0"]:
    d=json.loads(open("gpurun_out/s2/%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], json.dumps(d.get("roofline"))[:300])
    print(json.dumps({k:v for k,v in d.items() if k not in ("config","roofline","cpu_baseline")})[:2500])
PY
