# GPU box: extra SQ counter passes of the default bench step (each its own run, no trace domains), per-kernel ratios
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s2/pmcx; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_IFETCH SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES"; do
  i=$((i+1)); rm -rf $O/p$i
  timeout -k 10 300 rocprofv3 --pmc $set -d $O/p$i -o r --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $O/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for p in sorted(glob.glob("$O/p*/r_counter_collection.csv")):
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]; k = k[k.find("conv_"):] if "conv_" in k else k[k.rfind("::") + 2:]; k = k[:46]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); seen[k].add(r["Dispatch_Id"])
    for k, d in seen.items(): n[k] = len(d)
names = ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_SCA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_LDS_CMD_FIFO_FULL", "SQ_LDS_DATA_FIFO_FULL", "SQ_VMEM_TA_ADDR_FIFO_FULL", "SQ_VMEM_WR_TA_DATA_FIFO_FULL", "SQ_INST_LEVEL_LDS", "SQ_INST_LEVEL_VMEM", "SQ_IFETCH"]
print("kernel | " + " | ".join(x.replace("SQ_", "") for x in names) + "   (percent of SQ_WAVE_CYCLES)")
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    wc = c.get("SQ_WAVE_CYCLES", 0)
    if wc <= 0 or "conv" not in k: continue
    print(k, "|", " | ".join("%.1f" % (100 * c.get(x, 0) / wc) for x in names))
PY
