# GPU box: host rates again (staging pre-allocated), then conv_w4 stage stamps at C = 128 and C = 256
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_stamps; mkdir -p $O; cd $R
timeout -k 10 300 python tools/host_path_rate.py > $O/rate.log 2>&1; grep -v "^$" $O/rate.log | tail -8
for v in 128 256 128r; do echo "== stamps $v"; bash tools/s2_stamps.sh $v > $O/stamps_$v.log 2>&1; head -60 $O/stamps_$v.log; done
python -m image_restoration_platform_amd.build > /dev/null 2>&1
