# GPU box: conv_pk: restore + tiled tests, bench A/B against conv_w4 (IRE_PK=0), then its stage stamps
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_pk2; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_restore_gpu.py tests/test_tiled_gpu.py -x -q > $O/pytest.log 2>&1; rc=$?; tail -8 $O/pytest.log; [ $rc -eq 0 ] || exit $rc
for pk in 1 0 1 0; do
  IRE_PK=$pk timeout -k 10 200 python bench.py --no-cpu-baseline > $O/bench_pk$pk.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/bench_pk$pk.json").read().strip().splitlines()[-1])
print("IRE_PK=$pk", round(d["value"],1), "img/s;", {g["group"]: round(g["us_per_launch"], 1) for g in d["roofline"].get("per_group", []) if g["group"][:2] in ("L2", "L3")})
PY
done
bash tools/r04_pkstamps.sh 128 128r 2>&1 | grep -v "^wg 1" | head -64
