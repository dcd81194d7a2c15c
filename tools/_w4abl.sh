cd /tmp && export TMPDIR=/tmp
for d in 0 1 2 8 32 40 42; do
  IRE_W4=1 IRE_W4_DBG=$d timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/abl_w4_$d -o r --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > $GRAFT_REPO_ROOT/gpurun_out/abl_w4_$d.log 2>&1 || exit 1
done
