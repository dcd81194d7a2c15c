# GPU box: classifier parity suite, then the classify workload (kernel time per 8 x 1024^2 batch) and the default bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_cls; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_classifier_gpu.py tests/test_restore_gpu.py -k "classif or conditioning or golden" -x -q -m gpu > $O/tests.log 2>&1; tail -2 $O/tests.log
for rep in 1 2; do
timeout -k 10 200 python bench.py --workload classify --steps 50 --no-cpu-baseline > $O/cls.json 2> $O/cls.err || { tail -5 $O/cls.err; exit 1; }
python - <<PY
import json
d=json.loads(open("$O/cls.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("classify", round(d["value"]), "img/s; kernel family", round(r["algorithmic_bytes_per_step"] / r["achieved"] / 1e3, 1), "us per step; frac", round(r["frac"], 4))
PY
done
