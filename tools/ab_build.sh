# same-box A/B of BUILD-time variants of conv_rb.hip: rebuilds on the GPU box between runs
set -e
cd $GRAFT_REPO_ROOT
for v in $AB_VALUES; do
  env $AB_VAR=$v python -m image_restoration_platform_amd.build --force > /dev/null 2>&1
  for rep in 1 2; do
    timeout -k 10 200 python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$AB_VAR=$v', $rep, round(j['value'],1), round(j['ms_per_step'],3))"
  done
done
