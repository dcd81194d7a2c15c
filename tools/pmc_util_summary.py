"""Summarise the four `rocprofv3 --pmc` passes of tools/pmc_util.sh into profiles/<name>.md (per kernel, averaged per launch)."""
import collections, csv, glob, sys

src, dst = sys.argv[1], sys.argv[2]
title = sys.argv[3] if len(sys.argv) > 3 else "Round 1"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
n = collections.defaultdict(int)
for p in sorted(glob.glob(src + "/p*/r_counter_collection.csv")):
    seen = collections.defaultdict(dict)
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        k = k[k.find("conv_"):] if "conv_" in k else k[k.rfind("::") + 2:]
        k = k[:72]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        seen[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if p.endswith("p1/r_counter_collection.csv"):
        for k, d in seen.items():
            n[k] = len(d)
            dur[k] = sum(d.values()) / len(d) / 1e3
rows = []
for k, c in agg.items():
    if not n.get(k) or c.get("GRBM_GUI_ACTIVE", 0) == 0:
        continue
    cyc = c["GRBM_GUI_ACTIVE"] / n[k] / 8.0                       # summed over the 8 XCDs
    mf = max(c.get("SQ_INSTS_MFMA", 0.0), 1.0)
    rows.append((c["GRBM_GUI_ACTIVE"], k, n[k], dur[k], cyc / dur[k] / 1e3,
                 100.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / n[k] / (cyc * 1024.0),
                 c.get("SQ_INSTS_VALU", 0) / mf if c.get("SQ_INSTS_MFMA") else float("nan"),
                 c.get("SQ_INSTS_SALU", 0) / mf if c.get("SQ_INSTS_MFMA") else float("nan"),
                 c.get("SQ_INSTS_LDS", 0) / mf if c.get("SQ_INSTS_MFMA") else float("nan"),
                 100.0 * c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1),
                 100.0 * c.get("SQ_WAIT_INST_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1)))
with open(dst, "w") as f:
    f.write("# " + title + " — PMC counters per kernel, default bench step (1024x1024 bs 8)\n\n"
            "four separate `rocprofv3 --pmc` passes of `python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile`\n"
            "(2 steps traced; no trace domain combined with `--pmc`).  `MfmaUtil` = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs):\n"
            "the counter equals #MFMA x 32 cycles exactly (checked against the launch's FLOPs), so this is the achieved fraction of the MFMA\n"
            "pipe AT THE CLOCK THE KERNEL RAN AT (column GHz; profiled passes run a few % slower than unprofiled ones).\n"
            "VALU / SALU / LDS : MFMA are instruction-count ratios (wave-level); `LDS conflict` = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE;\n"
            "`wait` = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES.\n\n"
            "| kernel | launches | us | GHz | MfmaUtil % | VALU:MFMA | SALU:MFMA | LDS:MFMA | LDS conflict % | wait % |\n|---|---|---|---|---|---|---|---|---|---|\n")
    for r in sorted(rows, reverse=True):
        f.write("| `%s` | %d | %.1f | %.2f | %.1f | %.1f | %.1f | %.2f | %.1f | %.0f |\n" % r[1:])
print(open(dst).read()[-2500:])
