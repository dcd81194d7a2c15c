#!/usr/bin/env python3
"""Known-answer vectors for the classifier, derived ANALYTICALLY from the reference's own test
fixtures (server-node/tests/utils/imageFixtures.js:5-45): flat (180,180,180), dark (10,10,10),
colour cast (220,80,40), plus a seeded replacement for the unseeded Math.random() noise fixture
(imageFixtures.js:21-37 -> numpy default_rng(1234)).  The reference cannot run here (Node 12, no
sharp: SURVEY.md 8c), so the expected values are the closed forms worked out in SURVEY.md 8(c)
from classifier.js, NOT outputs of this repo's code.  Inputs are raw RGB (no JPEG round trip).
"""
import json
import os

KAT = [
    {"name": "clean_flat_180", "size": [128, 128], "fill": [180, 180, 180], "is_jpeg": True,
     # Laplacian of a flat field = 0 => var 0 => blur = 1; high-pass = pixel => var 0 => noise 0
     # mean 180/255 = .706 > .3 => lowLight 0; fade = (1-0)*.6 + (1-0)*.4 = 1; all means equal => colorShift 0
     "expect": {"blur": 1.0, "noise": 0.0, "lowLight": 0.0, "compression": 0.0, "scratch": 0.0, "fade": 1.0,
                "colorShift": 0.0}, "label": "blur"},
    {"name": "dark_10", "size": [128, 128], "fill": [10, 10, 10], "is_jpeg": True,
     # lowLight = (0.3 - 10/255) * 2
     "expect": {"blur": 1.0, "noise": 0.0, "lowLight": (0.3 - 10.0 / 255.0) * 2.0, "compression": 0.0,
                "scratch": 0.0, "fade": 1.0, "colorShift": 0.0}, "label": "blur"},
    {"name": "cast_220_80_40", "size": [128, 128], "fill": [220, 80, 40], "is_jpeg": True,
     # avg = 113.33; max dev = |220-113.33|/113.33 = .941 => min(1.882, 1) = 1
     "expect": {"blur": 1.0, "noise": 0.0, "lowLight": 0.0, "compression": 0.0, "scratch": 0.0, "fade": 1.0,
                "colorShift": 1.0}, "label": "blur"},
    {"name": "flat_png_not_jpeg", "size": [64, 40], "fill": [180, 180, 180], "is_jpeg": False,
     "expect": {"blur": 1.0, "noise": 0.0, "lowLight": 0.0, "compression": 0.0, "scratch": 0.0, "fade": 1.0,
                "colorShift": 0.0}, "label": "blur"},
    {"name": "noisy_seed1234", "size": [128, 128], "rng_seed": 1234, "is_jpeg": True,
     # uniform bytes: sigma of the clipped high-pass >> 50 => noise = 1; only inequalities are analytic here
     "expect_min": {"noise": 1.0}, "expect_max": {"blur": 0.0, "lowLight": 0.0}},
]

if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "classifier_kat.json")
    with open(out, "w") as f:
        json.dump(KAT, f, indent=1)
    print("wrote", out)
