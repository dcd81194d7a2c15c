# GPU box: workgroup timelines of conv_w4 (diagnostic build -DIRE_W4_TL), C = 128 / 256, RB1 / RB2; then the product build again.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_tl; mkdir -p $O; cd $R
env IRE_W4_TL=1 python -m image_restoration_platform_amd.build > $O/build.log 2>&1 || { tail $O/build.log; exit 1; }
for v in ${TLV:-128 128r 256 256r}; do
  env IRE_RB_STAMPS=$v IRE_W4_TL=$O/tl_$v.csv timeout -k 10 200 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-host-path --no-secondary > $O/bench_$v.json 2> $O/bench_$v.err || { tail -5 $O/bench_$v.err; exit 1; }
  echo "== $v"; python tools/r04_tl.py $O/tl_$v.csv | tee $O/tl_$v.txt
  python - <<PY
import json
d=json.loads(open("$O/bench_$v.json").read().strip().splitlines()[-1])
print("bench", d["value"], "img/s;", {g["group"]: round(g["us_per_launch"], 1) for g in d["roofline"].get("per_group", []) if g["group"][:2] in ("L2", "L3")})
PY
done
python -m image_restoration_platform_amd.build > /dev/null 2>&1
