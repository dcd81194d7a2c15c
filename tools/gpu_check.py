#!/usr/bin/env python3
"""Diagnostic run on the GPU box: classifier parity + per-layer RestoreNet comparison against the
CPU oracle (test infrastructure).  Prints detailed diffs; tests/ holds the asserting versions."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from image_restoration_platform_amd import synth, weights as W  # noqa: E402
from image_restoration_platform_amd.engine import Engine  # noqa: E402
from oracle import classifier as oc  # noqa: E402
from oracle import restorenet as onet  # noqa: E402


def main():
    h = int(os.environ.get("H", 64)); w = int(os.environ.get("WID", 96)); n = int(os.environ.get("N", 2))
    eng = Engine(max_batch=8, num_streams=1)
    imgs = synth.batch(n, h, w)
    sc, lb = eng.classify(imgs, is_jpeg=True)
    sums = eng.classifier_sums(n)
    ok = True
    for i in range(n):
        s_o, l_o, su = oc.classify(imgs[i], True, with_sums=True)
        same = np.array_equal(s_o.view(np.uint64), sc[i].view(np.uint64))
        print(f"classify img{i}: bit-exact={same} label gpu={lb[i]} cpu={l_o}")
        if not same:
            ok = False
            print("  gpu", sc[i]); print("  cpu", s_o)
            print("  sums gpu", sums[i].tolist()); print("  sums cpu", su.as_list())
    wts = W.generate(0)
    eng.debug_capture(True)
    out = eng.restore(imgs, scores=sc)
    cap = {}
    ref = onet.restore(imgs, sc, wts, emulate_bf16=True, capture=cap)
    names = ["stem"]
    for l in range(4):
        names += [f"enc{l}.rb0.h", f"enc{l}.rb0", f"enc{l}.rb1.h", f"enc{l}.rb1"]
        if l < 3:
            names.append(f"down{l}")
    names += ["mid.rb0.h", "mid.rb0", "mid.rb1.h", "mid.rb1"]
    for l in (2, 1, 0):
        names += [f"up{l}", f"fuse{l}", f"dec{l}.rb0.h", f"dec{l}.rb0", f"dec{l}.rb1.h", f"dec{l}.rb1"]
    for nm in names:
        try:
            a = eng.activation(nm)
        except Exception as e:  # noqa: BLE001
            print(f"{nm:14s} MISSING ({e})"); ok = False; continue
        r = cap[nm].reshape(-1)
        if a.size != r.size:
            print(f"{nm:14s} size {a.size} vs {r.size}"); ok = False; continue
        d = np.abs(a - r); sc_ = np.abs(r).mean() + 1e-9
        bad = int((d > 0.05 * (np.abs(r) + sc_)).sum())
        print(f"{nm:14s} max|d|={d.max():9.4f} mean|d|={d.mean():9.5f} mean|ref|={sc_:8.4f} rel={d.mean()/sc_:8.5f} outliers={bad}")
        if d.mean() / sc_ > 0.02:
            ok = False
            idx = int(np.argmax(d)); shp = cap[nm].shape
            print("   worst at", np.unravel_index(idx, shp), "gpu", a[idx], "ref", r[idx])
    d = np.abs(out.astype(np.int32) - ref.astype(np.int32))
    print(f"final u8 (vs bf16-emulating oracle): max={d.max()} mean={d.mean():.4f} frac>1={np.mean(d > 1):.5f}")
    ref32 = onet.restore(imgs, sc, wts)
    d32 = np.abs(out.astype(np.int32) - ref32.astype(np.int32))
    mse = np.mean((out.astype(np.float64) - ref32.astype(np.float64)) ** 2)
    print(f"final u8 (vs fp32 oracle): max={d32.max()} mean={d32.mean():.4f} frac>2={np.mean(d32 > 2):.5f} "
          f"psnr={10 * np.log10(255.0 ** 2 / max(mse, 1e-12)):.2f} dB; mean|out-in|={np.abs(ref32.astype(int) - imgs.astype(int)).mean():.2f}")
    eng.debug_capture(False)
    t0 = time.time(); eng.restore(imgs, scores=sc); print("restore wall s", time.time() - t0)
    print("GPU_CHECK", "OK" if ok else "FAIL")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
