# GPU box: same-box A/B of extra compiler flags over the whole library: bash tools/s2_xflags.sh "<flags A>" "<flags B>" ...   ("-" = none)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s2; mkdir -p $O
for rep in 1 2; do
for f in "$@"; do
  if [ "$f" = "-" ]; then x=""; else x="$f"; fi
  cd $R; env IRE_XFLAGS="$x" python -m image_restoration_platform_amd.build > $O/xf_build.log 2>&1 || { echo "[$f] build failed"; tail -3 $O/xf_build.log; continue; }
  if [ $rep = 1 ]; then timeout -k 10 600 python -m pytest tests/test_restore_gpu.py -x -q -m gpu -k "end_to_end or golden" > $O/xf_tests.log 2>&1; tail -1 $O/xf_tests.log; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-profile > $O/xf.json 2>> $O/xf_err.log || { echo "[$f] bench failed"; continue; }
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/s2/xf.json").read().strip().splitlines()[-1])
print("[$f]", round(d["value"],1), "img/s")
PY
done; done
cd $R; python -m image_restoration_platform_amd.build > /dev/null 2>&1
