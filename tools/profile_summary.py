"""Turn rocprofv3 outputs into the summaries committed under profiles/.

  python tools/profile_summary.py stats  <kernel_stats.csv> <out.md> "<title>" ["<footer>"]
  python tools/profile_summary.py traffic <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> <steps>

`traffic`: FETCH_SIZE and WRITE_SIZE come from two separate `rocprofv3 --pmc` passes of
`python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile` (2 steps traced).  Units are KiB; on gfx950
FETCH_SIZE reports half of the bytes of wide coalesced reads and is doubled (MI355X_MICROARCH.md, HBM / rocprofv3 section).
The conv3x3 family = every 3x3 conv kernel the engine's profiler counts in that family (conv_rb without the HEAD
instantiation, conv_w4 / conv_f8, conv_up -- which since round 2 carries the composed 1x1 fuse -- conv_down, and the 3x3
conv_mfma instantiations: TAPS == 9 with KC8 == 4, i.e. not the stem).
"""
import csv
import json
import sys
from collections import defaultdict


def stats(src, dst, title, footer=""):
    rows = list(csv.DictReader(open(src)))
    with open(dst, "w") as f:
        f.write(f"# {title}\n\n| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
        for r in rows:
            f.write(f"| `{r['Name'][:112]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |\n")
        if footer:
            f.write("\n" + footer + "\n")


def is_conv3x3(name):
    if "conv_rb_kernel" in name or "conv_pc_kernel" in name:   # the HEAD instantiation (last template argument true) is its own family
        return not name.rstrip().endswith(", true>(ire::ConvArgs)")
    if "conv_w4_kernel" in name or "conv_up_kernel" in name or "conv_down_kernel" in name or "conv_f8_kernel" in name or "conv_pk_kernel" in name or "conv_upq_kernel" in name or "conv_dnq_kernel" in name:
        return True
    if "conv_mfma_kernel<4, 9," in name and not name.rstrip().endswith("true>(ire::ConvArgs)"):   # last arg = HEAD: its own family
        return True
    return False


def _sum(path, counter):
    per = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        per[k][0] += float(r["Counter_Value"])
        per[k][1] += 1
    return per


def group_of(name, k):
    """Layer group (bench.py roofline.per_group key) of the k-th dispatch (0-based, within one step) of kernel `name`.
    The step's op order is fixed (engine.cpp::build_program): enc0..3, mid, dec2..0; the C >= 128 kernels serve level 2
    (enc2, dec2) and level 3 (enc3, mid) with ONE instantiation each, so the position in the step tells the level."""
    n = name.rstrip()
    if "conv_pk_kernel<" in n:          # conv_pk_kernel<C, RESID>: the width is the level
        args = n[n.find("<") + 1:n.find(">")].replace(" ", "").split(",")
        return "L%d.rb%d" % ({"128": 2, "256": 3}[args[0]], 2 if args[1] == "true" else 1)
    if "conv_w4_kernel" in n or "conv_f8_kernel" in n:
        resid = "<128, 8, true" in n or n.startswith("void ire::(anonymous namespace)::conv_f8_kernel<true") or "conv_f8_kernel<true" in n
        lvl = [2, 2, 3, 3, 3, 3, 2, 2][k % 8]
        return "L%d.rb%d" % (lvl, 2 if resid else 1)
    if "conv_pc_kernel<" in n or "conv_rb_kernel<" in n:
        args = n[n.find("<") + 1:n.find(">")].replace(" ", "").split(",")
        if args[-1] == "true":
            return "head"
        lvl = {"32": 0, "64": 1}.get(args[0])
        if lvl is None:
            return None
        return "L%d.rb%d" % (lvl, 2 if args[1] == "true" else 1)
    if "conv_dnq_kernel<" in n:         # conv_dnq_kernel<COUT>: the width is the level
        return {"128": "down1", "256": "down2"}.get(n[n.find("<") + 1:n.find(">")])
    if "conv_down_kernel" in n:         # default schedule: only down0 (IRE_DNQ=0: all three, in level order)
        return "down0"
    if "conv_up_kernel<" in n:
        return {"2": "up0", "4": "up1", "8": "up2"}.get(n[n.find("<") + 1:n.find(">")])
    if "conv_upq_kernel" in n:
        return "up2"
    if "conv_stem_kernel" in n:
        return "stem"
    return None


def _per_dispatch(path, counter):
    per = defaultdict(list)       # kernel -> [(dispatch id, value)]
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            per[r["Kernel_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return {k: [v for _, v in sorted(rows)] for k, rows in per.items()}


def traffic(fetch_csv, write_csv, dst, steps):
    steps = int(steps)
    fe, wr = _sum(fetch_csv, "FETCH_SIZE"), _sum(write_csv, "WRITE_SIZE")
    out = {
        "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile",
        "unit_note": "counters are KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); WRITE_SIZE as is",
        "family": "conv3x3 (conv_pc_kernel / conv_rb_kernel without their HEAD instantiation, conv_w4_kernel, conv_f8_kernel, conv_up_kernel, conv_upq_kernel, conv_dnq_kernel, conv_down_kernel; the v1 stride-2 conv_mfma_kernel instantiation where a switch selects it)",
        "per_kernel": {},
    }
    tot_f = tot_w = 0.0
    launches = 0
    for k in sorted(fe):
        if not is_conv3x3(k):
            continue
        f = fe[k][0] * 1024.0 * 2.0
        w = wr.get(k, [0.0, 0])[0] * 1024.0
        n = fe[k][1]
        tot_f += f
        tot_w += w
        launches += n
        out["per_kernel"][k[k.find("conv_"):][:100]] = {"launches": n // steps, "fetch_MiB_corrected": round(f / n / 2 ** 20, 1), "write_MiB": round(w / n / 2 ** 20, 1)}
    # per layer group (the same counters, dispatch by dispatch in launch order)
    fd, wd = _per_dispatch(fetch_csv, "FETCH_SIZE"), _per_dispatch(write_csv, "WRITE_SIZE")
    pg = defaultdict(lambda: [0.0, 0.0, 0])
    for k, vals in fd.items():
        per_step = len(vals) // steps
        wv = wd.get(k, [0.0] * len(vals))
        for i, v in enumerate(vals):
            g = group_of(k, i % max(1, per_step))
            if g is None:
                continue
            pg[g][0] += v * 1024.0 * 2.0
            pg[g][1] += (wv[i] if i < len(wv) else 0.0) * 1024.0
            pg[g][2] += 1
    out["per_group"] = {g: {"launches": n // steps, "fetch_MiB_corrected": round(f / n / 2 ** 20, 1), "write_MiB": round(w / n / 2 ** 20, 1),
                            "hbm_bytes_per_launch": (f + w) / n} for g, (f, w, n) in sorted(pg.items())}
    out["launches_per_step"] = launches // steps
    out["fetch_bytes_per_launch_corrected"] = tot_f / launches
    out["write_bytes_per_launch"] = tot_w / launches
    out["hbm_bytes_per_launch"] = (tot_f + tot_w) / launches
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("launches_per_step", "hbm_bytes_per_launch")}))


if __name__ == "__main__":
    {"stats": stats, "traffic": traffic}[sys.argv[1]](*sys.argv[2:])
