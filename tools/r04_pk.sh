# GPU box: host info + batcher rates with / without the CPU plan; then conv_pk: restore + tiled tests, bench A/B against conv_w4
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_pk; mkdir -p $O; cd $R
python - > $O/host.txt 2>&1 <<PY
import os, glob
print("nproc", os.cpu_count(), "allowed", sorted(os.sched_getaffinity(0)))
for n in sorted(glob.glob("/sys/devices/system/node/node*/cpulist")): print(n, open(n).read().strip())
from image_restoration_platform_amd.engine import Engine
e = Engine(max_batch=8); print("engine affinity", e.affinity()); e.close()
PY
cat $O/host.txt
echo "== host path, default"; timeout -k 10 300 python tools/host_path_rate.py 2>&1 | grep -v "^$" | head -7 | tee $O/rate_default.txt
echo "== host path, IRE_CPU_AFFINITY=off"; IRE_CPU_AFFINITY=off timeout -k 10 300 python tools/host_path_rate.py 2>&1 | grep -v "^$" | head -7 | tee $O/rate_off.txt
timeout -k 10 900 python -m pytest tests/test_restore_gpu.py tests/test_tiled_gpu.py -x -q > $O/pytest.log 2>&1; rc=$?; tail -8 $O/pytest.log; [ $rc -eq 0 ] || exit $rc
for pk in 1 0 1 0; do
  IRE_PK=$pk timeout -k 10 200 python bench.py --no-cpu-baseline > $O/bench_pk$pk.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/bench_pk$pk.json").read().strip().splitlines()[-1])
print("IRE_PK=$pk", round(d["value"],1), "img/s;", {g["group"]: round(g["us_per_launch"], 1) for g in d["roofline"].get("per_group", []) if g["group"][:2] in ("L2", "L3")})
PY
done
