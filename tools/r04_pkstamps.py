"""conv_pk stage stamps (diagnostic build PK_TICKS=1): consumer wave 0 and producer wave 8 of the first 8 workgroups."""
import sys
import numpy as np
v = np.array([int(x) for x in open(sys.argv[1]).read().split()], dtype=np.uint64).astype(np.int64)
a = v[: 8 * 2 * 32 * 4].reshape(8, 2, 32, 4)
for wg in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2):
    c, p = a[wg, 0], a[wg, 1]
    print("wg", wg)
    print("  stage | consumer: k-loop  barrier+next-start  (epilogue) | producer: wait-landed  produce  wait-slab  barrier-wait | producer start - consumer start")
    for s in range(24):
        if c[s, 0] == 0 or c[s + 1, 0] == 0: break
        kl = c[s, 1] - c[s, 0]
        epi = c[s, 2] - c[s, 1] if c[s, 2] else 0
        nxt = c[s + 1, 0] - (c[s, 2] if c[s, 2] else c[s, 1])
        pw = p[s, 1] - p[s, 0]; pp = p[s, 2] - p[s, 1]; ps = p[s, 3] - p[s, 2]; pb = (p[s + 1, 0] - p[s, 3]) if p[s + 1, 0] else 0
        print("  %5d | %8d %8d %10s | %8d %8d %8d %8d | %8d   stage total %d" % (s, kl, nxt, ("%d" % epi) if epi else "-", pw, pp, ps, pb, p[s, 0] - c[s, 0], c[s + 1, 0] - c[s, 0]))

w = v[4096: 4096 + 16 * 12 * 2].reshape(16, 12, 2)
if w.any():
    print("workgroup 0: arrival at the stage barrier per wave, ticks after the stage's first arrival (release - first arrival in the last column)")
    print("  stage | consumers 0..7" + " " * 58 + "| producers 8..11")
    for s_ in range(16):
        if not w[s_, :, 0].all(): break
        t0 = w[s_, :, 0].min()
        arr = w[s_, :, 0] - t0
        print("  %5d | %s | %s | released +%d ; previous release -> first arrival %s" % (s_, " ".join("%6d" % x for x in arr[:8]), " ".join("%6d" % x for x in arr[8:]),
              w[s_, :, 1].min() - t0, (t0 - w[s_ - 1, :, 1].min()) if s_ else "-"))
