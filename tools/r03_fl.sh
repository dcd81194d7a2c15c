# GPU box: fusion chain by FUSE_FL (loads in flight in the SAD kernels' last-workgroup row sum).   usage: bash tools/r03_fl.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_fl; mkdir -p $O; cd $R
for v in 16 24 32 16 24 32; do
  touch image_restoration_platform_amd/csrc/fusion.hip
  env FUSE_FL=$v python -m image_restoration_platform_amd.build > /dev/null 2>&1 || { echo build failed; exit 1; }
  timeout -k 10 200 python bench.py --workload fusion --steps 10 --no-cpu-baseline > $O/f_$v.json 2>/dev/null
  python3 -c "
import json; d=json.loads(open('$O/f_$v.json').read().strip().splitlines()[-1]); b=d['roofline']['batched_entry']; print('FL $v: chain %.1f us frac %.4f | single-set family frac %.4f' % (1e3*b['kernel_chain_ms'], b['frac'], d['roofline']['frac']))"
done
touch image_restoration_platform_amd/csrc/fusion.hip; python -m image_restoration_platform_amd.build > /dev/null 2>&1
