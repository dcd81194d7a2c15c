# GPU box: two lanes (HIP streams) with persistent grids of all / half the CUs, unprofiled (overlapping launches skew per-kernel events)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_lanes; mkdir -p $O; cd $R
run() { # label, env..., args
  label=$1; shift
  for rep in 1 2; do
    env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-host-path --no-secondary --no-profile $BARGS > $O/b.json 2> $O/b.err || { tail -3 $O/b.err; exit 1; }
    python - <<PY
import json
d=json.loads(open("$O/b.json").read().strip().splitlines()[-1])
print("$label", round(d["value"],1), "img/s", round(d["ms_per_step"],3), "ms")
PY
  done
}
BARGS="--streams 1" run "1 lane, 256 CUs" X=1
BARGS="--streams 2" run "2 lanes, 256 CUs" X=1
BARGS="--streams 2" run "2 lanes, 128 CUs" IRE_GRID_CUS=128
BARGS="--streams 2" run "2 lanes, 160 CUs" IRE_GRID_CUS=160
BARGS="--streams 4" run "4 lanes, 64 CUs" IRE_GRID_CUS=64
BARGS="--streams 1" run "1 lane, 128 CUs" IRE_GRID_CUS=128
timeout -k 10 300 python -m pytest tests/test_restore_gpu.py -x -q -k "stream or invariance or batch" 2>&1 | tail -2
