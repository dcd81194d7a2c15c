# GPU box: round 4's profile set -- default bench (per-level roofline + host_path + cpu_baseline), the secondary workloads
# (512^2, cfg 3, cfg 4 bf16 / fp8, classifier), kernel-trace stats, FETCH/WRITE PMC passes, utilisation PMC passes.
# Every pass is its own run; --pmc is never combined with a trace domain.   usage: bash tools/r04_profile.sh [tag]
set -e
TAG=${1:-v1}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_$TAG; mkdir -p $O
cd $R
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout -k 10 200 python bench.py --no-profile --no-cpu-baseline --no-host-path > $O/bench_noprofile.json 2>> $O/bench_default.err
timeout -k 10 200 python bench.py --size 512 --no-cpu-baseline --no-host-path --no-secondary > $O/bench_512.json 2>> $O/bench_default.err
timeout -k 10 200 python bench.py --workload tiled --steps 10 --no-cpu-baseline > $O/bench_tiled_bf16.json 2>> $O/bench_default.err
timeout -k 10 200 python bench.py --workload tiled --precision fp8 --steps 10 --no-cpu-baseline > $O/bench_tiled_fp8.json 2>> $O/bench_default.err
timeout -k 10 200 python bench.py --workload fusion --steps 10 --no-cpu-baseline > $O/bench_fusion.json 2>> $O/bench_default.err
timeout -k 10 200 python bench.py --workload classify --steps 50 --no-cpu-baseline > $O/bench_classify.json 2>> $O/bench_default.err
echo "benches done"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o r --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path --no-secondary > $O/stats.log 2>&1
echo "stats done"
# kernel-trace rows of the secondary workloads (classifier, fusion incl. the batched entry, tiled bf16 / fp8, 512^2)
for wl in "classify --steps 30" "fusion --steps 10" "tiled --steps 6" "tiled --precision fp8 --steps 6"; do
  tag=$(echo $wl | tr -d ' -' | cut -c1-16)
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_$tag -o r --output-format csv -- python3 $R/bench.py --workload $wl --warmup 2 --no-cpu-baseline > $O/stats_$tag.log 2>&1 || echo "stats $tag failed"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_512 -o r --output-format csv -- python3 $R/bench.py --size 512 --steps 8 --warmup 2 --no-cpu-baseline --no-host-path > $O/stats_512.log 2>&1 || echo "stats 512 failed"
echo "secondary stats done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o r --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-path --no-secondary --no-profile > $O/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o r --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-path --no-secondary --no-profile > $O/pmc_write.log 2>&1
echo "traffic done"
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set -d $O/pmc/p$i -o r --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-path --no-secondary --no-profile > $O/pmc_p$i.log 2>&1 || echo "pass $i failed"
  echo "pmc pass $i done"
done
for f in $O/bench_*.json; do echo $f; cut -c1-200 $f; done
