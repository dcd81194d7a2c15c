"""server-node side of the boundary: the N-API shim + JS adapters that implement the reference's two
duck-typed seams (classifier.analyze / geminiClient.restoreImage).  Runs with the Node that ships in
the image (v12: the reference's own sources need >= 18, the adapters are written to load on both)."""
import hashlib
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from image_restoration_platform_amd import synth, weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE_DIR = os.path.join(ROOT, "image_restoration_platform_amd", "node")
pytestmark = pytest.mark.skipif(shutil.which("node") is None, reason="node is not installed")


def _raw(img, is_jpeg=True):
    h, w, _ = img.shape
    return b"RAW1" + bytes([w & 255, w >> 8, h & 255, h >> 8, 1 if is_jpeg else 0]) + img.tobytes()


def _run(tmp_path, spec):
    p = tmp_path / "case.json"
    p.write_text(json.dumps(spec))
    r = subprocess.run(["node", os.path.join(NODE_DIR, "test_adapters.js"), str(p)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_addon_loads_and_fails_loudly_without_gpu(tmp_path):
    import torch
    assert os.path.exists(os.path.join(NODE_DIR, "ire_napi.node")), "run __graft_entry__.build()"
    img = tmp_path / "a.raw"
    img.write_bytes(_raw(synth.image(1, 64, 64)))
    out = _run(tmp_path, {"weights": weights.ensure_default(0), "image": str(img)})
    assert out["loaded"] is True
    if not torch.cuda.is_available():
        assert out["engine"] is False and "service unavailable" in out["initError"]


@pytest.mark.gpu
def test_node_seams_match_python_host(engine, tmp_path):
    img = synth.image(1, 72, 104)                      # not a multiple of 8 in H? 72 is; W=104 is: use ragged below
    img = np.ascontiguousarray(img[:70, :101])          # forces the adapter's replicate padding + crop
    f = tmp_path / "a.raw"
    f.write_bytes(_raw(img))
    views = synth.fusion_views(64, 64)
    vf = []
    for i, v in enumerate(views):
        p = tmp_path / f"v{i}.raw"
        p.write_bytes(_raw(v))
        vf.append(str(p))
    out = _run(tmp_path, {"weights": weights.ensure_default(0), "image": str(f), "fuse": vf})
    assert out["engine"] is True and out["allEqual"] is True and out["success"] is True
    scores, _ = engine.classify(img, is_jpeg=True)
    from image_restoration_platform_amd.prompt_enhancer import KEYS
    assert [out["scores"][k] for k in KEYS] == [float(x) for x in scores[0]]          # bit-exact through JSON
    padded = np.pad(img, ((0, 2), (0, 3), (0, 0)), mode="edge")
    ref = engine.restore(padded, scores=None, is_jpeg=True)[0][:70, :101]
    assert out["restoredSha"] == hashlib.sha256(np.ascontiguousarray(ref).tobytes()).hexdigest()
    assert out["metadata"]["estimatedCostUsd"] == 0 and out["metadata"]["billedTokens"] is None
    assert out["metadata"]["providerRequestId"].startswith("ire-")
    assert out["bad"]["success"] is False and out["bad"]["error"]["code"] == "RESTORATION_FAILED"
    assert out["fusedLen"] == 9 + 64 * 64 * 3
