# GPU box: conv_pk stage stamps (diagnostic build PK_TICKS=1): bash tools/r04_pkstamps.sh "<cout>[r]" ...
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_pkstamps; mkdir -p $O; cd $R
env PK_TICKS=1 python -m image_restoration_platform_amd.build > $O/build.log 2>&1 || { tail $O/build.log; exit 1; }
for v in "$@"; do
  env IRE_RB_STAMPS=$v IRE_STAMPS_RAW=$O/raw_$v.txt timeout -k 10 200 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile > $O/bench_$v.json 2> $O/bench_$v.err || { tail -5 $O/bench_$v.err; exit 1; }
  echo "== $v"; python tools/r04_pkstamps.py $O/raw_$v.txt 2 | tee $O/stamps_$v.txt
done
python -m image_restoration_platform_amd.build > /dev/null 2>&1
