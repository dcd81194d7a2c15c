# GPU box: conv_dnq build variants / timing ablations: bash tools/r04_dnqab.sh "<DQ defs A>" "<DQ defs B>" ...
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_dnqab; mkdir -p $O; cd $R
i=0
for defs in "$@"; do
  i=$((i+1))
  env DQ_DEFS="$defs" python -m image_restoration_platform_amd.build > $O/build_$i.log 2>&1 || { tail $O/build_$i.log; exit 1; }
  if [ -n "$DQ_TESTS" ]; then timeout -k 10 600 python -m pytest tests/test_restore_gpu.py tests/test_tiled_gpu.py -x -q -m gpu > $O/tests_$i.log 2>&1; tail -2 $O/tests_$i.log; fi
  for rep in 1 2; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-host-path --no-secondary > $O/bench_$i.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/bench_$i.json").read().strip().splitlines()[-1])
print("[$defs]", round(d["value"],1), "img/s;", {g["group"]: round(g["us_per_launch"], 1) for g in d["roofline"].get("per_group", []) if g["group"][:2] in ("do",)})
PY
  done
done
env DQ_DEFS="" python -m image_restoration_platform_amd.build > /dev/null 2>&1
