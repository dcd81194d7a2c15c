"""Workgroup timeline of one conv_w4 launch (diagnostic build -DIRE_W4_TL; tools/r04_tl.sh): where a launch's time goes
between kernel entry and exit, over all workgroups.  Input: the CSV the engine writes at teardown (IRE_W4_TL=<path>)."""
import csv
import sys
from collections import defaultdict

import numpy as np


def main(path):
    rows = defaultdict(dict)
    for r in csv.DictReader(open(path)):
        rows[int(r["wg"])][int(r["slot"])] = (int(r["realtime_10ns"]), int(r["memtime"]))
    wgs = sorted(w for w in rows if 0 in rows[w] and 12 in rows[w])
    if not wgs:
        print("no stamps"); return
    t0 = min(rows[w][0][0] for w in wgs)
    us = lambda w, s: (rows[w][s][0] - t0) / 100.0
    nit = max([k for w in wgs for k in rows[w] if 3 <= k <= 11] or [2]) - 2
    print("workgroups %d, items per workgroup %s" % (len(wgs), nit))
    def stat(name, v):
        v = np.asarray(v, dtype=float)
        print("  %-34s min %8.2f  p10 %8.2f  med %8.2f  p90 %8.2f  max %8.2f" % (name, v.min(), np.percentile(v, 10), np.median(v), np.percentile(v, 90), v.max()))
    print("absolute times (us after the first workgroup's entry):")
    stat("entry", [us(w, 0) for w in wgs])
    stat("gn_fold done", [us(w, 1) for w in wgs])
    stat("prologue done (first MFMA)", [us(w, 2) for w in wgs])
    for k in range(min(nit, 9)):
        stat("item %d done" % k, [us(w, 3 + k) for w in wgs if 3 + k in rows[w]])
    stat("exit", [us(w, 12) for w in wgs])
    print("durations (us):")
    stat("gn_fold", [us(w, 1) - us(w, 0) for w in wgs])
    stat("prologue", [us(w, 2) - us(w, 1) for w in wgs])
    for k in range(min(nit, 9)):
        stat("item %d" % k, [us(w, 3 + k) - us(w, 2 + k) for w in wgs if 3 + k in rows[w]])
    last = 3 + min(nit, 9) - 1
    stat("last item -> exit", [us(w, 12) - us(w, last) for w in wgs])
    stat("whole workgroup", [us(w, 12) - us(w, 0) for w in wgs])
    clk = [(rows[w][12][1] - rows[w][0][1]) / max(1, rows[w][12][0] - rows[w][0][0]) * 0.1 for w in wgs]
    stat("shader clock (GHz)", clk)
    ticks = lambda w, a, b: rows[w][b][1] - rows[w][a][1]
    print("ticks (shader cycles):")
    stat("gn_fold", [ticks(w, 0, 1) for w in wgs])
    stat("prologue", [ticks(w, 1, 2) for w in wgs])
    for k in range(min(nit, 9)):
        stat("item %d" % k, [ticks(w, 2 + k, 3 + k) for w in wgs if 3 + k in rows[w]])
    print("by XCC id: median exit, median whole")
    byx = defaultdict(list)
    for w in wgs:
        byx[int(rows[w].get(13, (0, 0))[0]) & 0xf].append(w)
    for x in sorted(byx):
        ws = byx[x]
        print("  xcc %d: n %3d  entry med %6.2f  exit med %7.2f max %7.2f  whole med %7.2f" % (
            x, len(ws), np.median([us(w, 0) for w in ws]), np.median([us(w, 12) for w in ws]), max(us(w, 12) for w in ws),
            np.median([us(w, 12) - us(w, 0) for w in ws])))


if __name__ == "__main__":
    main(sys.argv[1])
