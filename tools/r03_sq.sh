# GPU box: what the waves of each kernel spend their cycles on (SQ counters, quad-cycle units), two --pmc passes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_sq; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|TA_[A-Z_0-9]*\|TCP_[A-Z_0-9]*" | sort -u > $O/counters.txt; wc -l $O/counters.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set -d $O/p$i -o r --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-path --no-profile > $O/p$i.log 2>&1 || echo "pass $i failed: $(tail -2 $O/p$i.log)"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for p in sorted(glob.glob("$O/p*/**/r_counter_collection.csv", recursive=True)):
    seen=set()
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]; k = k[k.find("conv_"):][:48] if "conv_" in k else k[-40:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if p.endswith("p1/r_counter_collection.csv") or "/p1/" in p:
            if (k, r["Dispatch_Id"]) not in seen: seen.add((k, r["Dispatch_Id"])); n[k]+=1
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES",0)):
    wc = c.get("SQ_WAVE_CYCLES", 0)
    if wc <= 0: continue
    f = lambda x: 100.0*c.get(x,0)/wc
    print("%-50s n=%3d parked %4.1f%% issue-stall %4.1f%% active %4.1f%% (valu %4.1f lds %4.1f vmem %4.1f sca %4.1f) | insts/launch: valu %.0fk vmem_rd %.0fk vmem_wr %.0fk lds %.0fk | vmem_cyc %4.1f%% wait_lds %4.1f%% | TA_BUSY_sum/GUI %.2f TCP_pending/GUI %.2f" % (
        k, n[k], f("SQ_WAIT_ANY"), f("SQ_WAIT_INST_ANY"), f("SQ_ACTIVE_INST_ANY"), f("SQ_ACTIVE_INST_VALU"), f("SQ_ACTIVE_INST_LDS"), f("SQ_ACTIVE_INST_VMEM"), f("SQ_ACTIVE_INST_SCA"),
        c.get("SQ_INSTS_VALU",0)/max(n[k],1)/1e3, c.get("SQ_INSTS_VMEM_RD",0)/max(n[k],1)/1e3, c.get("SQ_INSTS_VMEM_WR",0)/max(n[k],1)/1e3, c.get("SQ_INSTS_LDS",0)/max(n[k],1)/1e3,
        f("SQ_INST_CYCLES_VMEM"), f("SQ_WAIT_INST_LDS"), c.get("TA_TA_BUSY_sum",0)/max(c.get("GRBM_GUI_ACTIVE",1),1), c.get("TCP_PENDING_STALL_CYCLES_sum",0)/max(c.get("GRBM_GUI_ACTIVE",1),1)))
PY
