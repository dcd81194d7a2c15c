R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_w4; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_restore_gpu.py tests/test_tiled_gpu.py -x -q -m gpu > $O/tests_nt64.log 2>&1 || { tail -30 $O/tests_nt64.log; exit 1; }
tail -3 $O/tests_nt64.log
for rep in 1 2; do for v in 1 0; do
  env IRE_W4_SPLIT=$v timeout -k 10 200 python bench.py --size 512 --steps 30 --no-cpu-baseline --no-host-path --no-profile | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('512^2 bs8 IRE_W4_SPLIT=$v', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms')"
done; done
for v in 1 0; do env IRE_W4_SPLIT=$v timeout -k 10 200 python bench.py --size 1024 --batch 2 --steps 30 --no-cpu-baseline --no-host-path --no-profile | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1024^2 bs2 IRE_W4_SPLIT=$v', round(d['value'],1), 'img/s')"; done
