# GPU box: 512^2 bs 8 (BASELINE cfg 1) with 1 / 2 / 4 lanes, unprofiled and profiled
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_lanes512; mkdir -p $O; cd $R
for rep in 1 2; do for st in 1 2 4; do
  timeout -k 10 200 python bench.py --size 512 --streams $st --no-cpu-baseline --no-host-path --no-secondary --no-profile > $O/b_$st.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/b_$st.json").read().strip().splitlines()[-1])
print("512^2 lanes=$st", round(d["value"],1), "img/s", round(d["ms_per_step"],3), "ms")
PY
done; done
