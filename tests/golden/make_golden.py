#!/usr/bin/env python3
"""Golden vectors generated HERE by this repo's CPU oracle (the Node reference cannot run in this
container: SURVEY.md 8c) -- they pin the oracle against drift and give the GPU tests fixed targets.

  classifier_golden.json : seeded synthetic images (image_restoration_platform_amd.synth) ->
                           7 scores as float64 hex, label, the 14 integer accumulators, top-3
                           issues and the enhanced prompt (promptEnhancer rules).
  restore_golden_*.npy   : fp32 RestoreNet-v0 oracle output (uint8) for two 64x64 seeded inputs.
Run: python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from image_restoration_platform_amd import synth, weights  # noqa: E402
from image_restoration_platform_amd.prompt_enhancer import KEYS, PromptEnhancerService, argmax_label, identify_top_issues  # noqa: E402
from oracle import classifier as oc  # noqa: E402
from oracle import restorenet as onet  # noqa: E402

CASES = [(i, h, w, jp) for i, (h, w, jp) in enumerate(
    [(64, 64, 1), (64, 96, 1), (37, 53, 1), (128, 128, 0), (96, 160, 1), (256, 256, 1), (17, 65, 1), (16, 64, 0),
     (200, 120, 1), (72, 136, 1)])]


def main():
    enh = PromptEnhancerService()
    out = []
    for i, h, w, jp in CASES:
        img = synth.image(i, h, w)
        s, l, sums = oc.classify(img, bool(jp), with_sums=True)
        deg = {k: float(s[j]) for j, k in enumerate(KEYS)}
        out.append({"index": i, "h": h, "w": w, "is_jpeg": jp, "scores_hex": [float(x).hex() for x in s],
                    "label": int(l), "label_name": argmax_label(deg), "sums": [int(x) for x in sums.as_list()],
                    "top_issues": [[t["type"], t["severity"]] for t in identify_top_issues(deg)],
                    "prompt": enh.enhance(deg, None)})
    with open(os.path.join(HERE, "classifier_golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    w0 = weights.generate(0)
    imgs = synth.batch(2, 64, 64, start=40)
    sc = np.stack([oc.classify(im, True)[0] for im in imgs])
    ref = onet.restore(imgs, sc, w0)
    np.save(os.path.join(HERE, "restore_golden_in.npy"), imgs)
    np.save(os.path.join(HERE, "restore_golden_out.npy"), ref)
    print("wrote", len(out), "classifier cases and restore goldens", ref.shape)


if __name__ == "__main__":
    main()
