# per-kernel MFMA / VALU / LDS counters of the default bench step (separate passes; no trace domains combined with --pmc)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set -d $O/p$i -o r --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $O/p$i.log 2>&1 || echo "pass $i failed"
done
ls $O
