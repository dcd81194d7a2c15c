# GPU box: restore / tiled / classifier suites on the build with the reworked gn_fold, three default benches, then the workgroup timelines at 1024^2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_check3; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_restore_gpu.py tests/test_tiled_gpu.py tests/test_node_adapter.py tests/test_serving.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -5 $O/pytest.log; [ $rc -eq 0 ] || exit $rc
for i in 1 2 3; do
timeout -k 10 300 python bench.py --no-cpu-baseline --no-host-path > $O/bench_$i.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python - <<PY
import json
d=json.loads(open("$O/bench_$i.json").read().strip().splitlines()[-1])
print("bench", round(d["value"],1), "img/s", round(d["ms_per_step"],3), "ms; frac", round(d["roofline"]["frac"],4), "per-layer", round(d["roofline"]["frac_of_per_layer_roofline"],4), "cfg1", round(d["secondary"]["cfg1_512"]["images_per_sec"],1), round(d["secondary"]["cfg1_512"]["conv3x3_family"]["frac"],4))
print({g["group"]: round(g["us_per_launch"], 1) for g in d["roofline"].get("per_group", [])})
PY
done
TLV="32 32r 64 64r 128 128r 256 256r" bash tools/r04_tl.sh > $O/tl.txt 2>&1; grep -E "^==|gn_fold  |prologue  |whole workgroup|exit  |shader clock" $O/tl.txt | grep -v "ticks" | head -90
