# GPU box: default bench three ways -- event timing of the dominant family (the default), of every family, and none.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s2; mkdir -p $O; cd $R
for m in "" "--profile-all" "--no-profile" ""; do
timeout -k 10 200 python bench.py --no-cpu-baseline $m > $O/evt.json 2>> $O/evt_err.log || exit 1
python - <<PY
import json
d=json.loads(open("gpurun_out/s2/evt.json").read().strip().splitlines()[-1])
r=d.get("roofline") or {}
print("mode [$m]:", round(d["value"],1), "img/s", round(d["ms_per_step"],3), "ms  frac", r.get("frac"), "launch_us", r.get("avg_launch_us"), r.get("launches_per_step"))
PY
done
