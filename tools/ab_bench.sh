# same-box A/B of engine switches: each "NAME ENV" line of $AB_LIST (default list below) is benchmarked twice, interleaved
# (boxes of the pool differ by several %, two runs on one box by < 1 %)
set -e
cd $GRAFT_REPO_ROOT
LIST="${AB_LIST:-default IRE_NOP=1;w4off IRE_W4=0;up64 IRE_UP_RB_MINC=64;slot IRE_SLOT_STATS=1;lanes2 IRE_STREAMS=2}"
for rep in 1 2; do
  echo "$LIST" | tr ';' '\n' | while read -r name envs; do
    [ -z "$name" ] && continue
    env $envs timeout -k 10 200 python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', $rep, round(j['value'],1), round(j['ms_per_step'],3))"
  done
done
