cd /tmp && export TMPDIR=/tmp
for wv in ${W4_WAVES:-8}; do for d in ${W4_DBGS:-0}; do
  IRE_W4=1 IRE_W4_WAVES=$wv IRE_W4_DBG=$d timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/abl_w4_${wv}_$d -o r --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > $GRAFT_REPO_ROOT/gpurun_out/abl_w4_${wv}_$d.log 2>&1 || exit 1
done; done
