# GPU box: conv_pk: restore + tiled tests, then bench per IRE_PK (2 = all convs, 1 = no-residual convs only, 0 = conv_w4) and producer priority, stamps
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_pk3; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_restore_gpu.py tests/test_tiled_gpu.py -x -q > $O/pytest.log 2>&1; rc=$?; tail -4 $O/pytest.log; [ $rc -eq 0 ] || exit $rc
for pk in 1 2 0 1 2 0; do
  IRE_PK=$pk timeout -k 10 200 python bench.py --no-cpu-baseline --no-host-path --no-secondary > $O/bench_pk$pk.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/bench_pk$pk.json").read().strip().splitlines()[-1])
print("IRE_PK=$pk", round(d["value"],1), "img/s;", {g["group"]: round(g["us_per_launch"], 1) for g in d["roofline"].get("per_group", []) if g["group"][:2] in ("L2", "L3")})
PY
done
bash tools/r04_pkstamps.sh 128 2>&1 | grep -v "^wg 1" | head -30
bash tools/r04_pkab.sh "-DPK_PRIO=2" "-DPK_PRIO=3" "-DPK_PRIO=0" "-DPK_CDMA=0"
