R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_pk5; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_restore_gpu.py -x -q -k "producer_consumer or alternate or seeded or end_to_end" > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; [ $rc -eq 0 ] || exit $rc
IRE_PK=0 timeout -k 10 200 python bench.py --no-cpu-baseline --no-host-path --no-secondary > $O/b0.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("$O/b0.json").read().strip().splitlines()[-1])
print("IRE_PK=0", round(d["value"],1), "img/s;", {g["group"]: round(g["us_per_launch"], 1) for g in d["roofline"].get("per_group", []) if g["group"][:2] in ("L2", "L3")})
PY
export IRE_PK=2
bash tools/r04_pkab.sh "-DPK_DBUF=1 -DPK_INTERIOR=1" "-DPK_DBUF=0 -DPK_INTERIOR=0" "-DPK_DBUF=0 -DPK_INTERIOR=1" "-DPK_DBUF=1 -DPK_INTERIOR=0" "-DPK_DBUF=0 -DPK_INTERIOR=0 -DPK_RDMA=0"
