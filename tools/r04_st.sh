# GPU box: output-store cache policy of conv_dnq / conv_upq (write-through sc1 = 16, write-back = 0, nt = 2): same-box A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_st; mkdir -p $O; cd $R
for rep in 1 2; do for st in 16 0 2; do
  env DQ_DEFS="-DDQ_ST=$st" UQ_DEFS="-DUQ_ST=$st" python -m image_restoration_platform_amd.build > $O/build.log 2>&1 || { tail $O/build.log; exit 1; }
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-host-path --no-secondary > $O/b_$st.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/b_$st.json").read().strip().splitlines()[-1])
print("ST=$st", round(d["value"],1), "img/s;", {g["group"]: round(g["us_per_launch"], 1) for g in d["roofline"].get("per_group", []) if g["group"] in ("down1", "down2", "up2", "L2.rb1", "L3.rb1")})
PY
done; done
env DQ_DEFS="" UQ_DEFS="" python -m image_restoration_platform_amd.build > /dev/null 2>&1
