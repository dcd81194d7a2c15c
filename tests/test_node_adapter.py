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
    f8 = tmp_path / "a8.raw"
    f8.write_bytes(_raw(synth.image(2, 72, 104)))
    f0 = tmp_path / "cfg0.raw"
    img0 = synth.image(3, 256, 256)
    f0.write_bytes(_raw(img0))
    out = _run(tmp_path, {"weights": weights.ensure_default(0), "image": str(f), "image8": str(f8), "fuse": vf, "worker": True, "concurrent": 8,
                          "cfg0": str(f0), "textResults": True})
    assert out["engine"] is True and out["allEqual"] is True and out["success"] is True
    scores, _ = engine.classify(img, is_jpeg=True)
    from image_restoration_platform_amd.prompt_enhancer import KEYS
    assert [out["scores"][k] for k in KEYS] == [float(x) for x in scores[0]]          # bit-exact through JSON
    padded = np.pad(img, ((0, 2), (0, 3), (0, 0)), mode="edge")
    ref = engine.restore(padded, scores=scores, is_jpeg=True)[0][:70, :101]       # conditioned on the image's own scores, not the padded copy's
    assert out["restoredSha"] == hashlib.sha256(np.ascontiguousarray(ref).tobytes()).hexdigest()
    assert out["metadata"]["estimatedCostUsd"] == 0 and out["metadata"]["billedTokens"] is None
    assert out["metadata"]["providerRequestId"].startswith("ire-")
    assert out["bad"]["success"] is False and out["bad"]["error"]["code"] == "RESTORATION_FAILED"
    assert out["fusedLen"] == 9 + 64 * 64 * 3
    # 8 concurrent Node jobs are served by the engine's batcher in shared batches (engine-side counters), with identical pixels
    cc = out["concurrent"]
    assert cc["images"] == 8 and cc["batches"] <= 2 and cc["allEqual"] and cc["sameAsSingle"], cc
    assert cc["health"]["ok"] is True and cc["health"]["info"]["status"] == "ok" and cc["health"]["info"]["imagesPerSec"] > 0
    assert out["once"] == {"scoresEqual": True, "pixelsEqual": True, "cached": True, "batches": 1}
    # BASELINE cfg 0 (one 256 x 256 job, default prompt, no fusion) through the Node seam == the Python host
    sc0, _ = engine.classify(img0, is_jpeg=True)
    assert out["cfg0"]["success"] is True and [out["cfg0"]["scores"][k] for k in KEYS] == [float(x) for x in sc0[0]]
    assert out["cfg0"]["sha"] == hashlib.sha256(np.ascontiguousarray(engine.restore(img0, scores=sc0, is_jpeg=True)[0]).tobytes()).hexdigest()
    # resultCodec 'png-device': restoreImage's base64Image is the device's text -- the oracle's PNG (stored deflate) of the same restored pixels
    from oracle import encode as oenc
    img8 = synth.image(2, 72, 104)
    sc8, _ = engine.classify(img8, is_jpeg=True)
    want = oenc.png_base64(engine.restore(img8, scores=sc8, is_jpeg=True)[0])
    assert out["textResult"]["head"] == "PNG" and out["textResult"]["chars"] == len(want)
    assert out["textResult"]["sha"] == hashlib.sha256(want).hexdigest()
    wk = out["worker"]                                   # the BullMQ-style worker over the engine-backed seams
    assert wk["good"]["status"] == "succeeded" and wk["good"]["providerRequestId"].startswith("ire-")
    assert wk["err"]["unrecoverable"] is True and wk["err"]["type"] == "INVALID_INPUT"
    assert wk["updates"] == ["running", "succeeded", "running"]


def test_queue_worker_contract():
    """restoration_worker.js against a miniature BullMQ Worker: queue defaults and back-off of jobQueue.js:4-9,37-45,
    retry / DLQ / refund / job-record rules of design.md:820-884,912-933."""
    r = subprocess.run(["node", os.path.join(NODE_DIR, "test_worker.js")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    o = json.loads(r.stdout.strip().splitlines()[-1])
    assert o["defaults"] == {"queueName": "image-restoration-jobs", "attempts": 5, "backoffBaseMs": 1000, "backoffJitter": 0.3,
                             "removeOnComplete": 100, "removeOnFail": 500, "deadLetterName": "image-restoration-dlq", "concurrency": 5}
    assert o["envDefaults"]["queueName"] == "q2" and o["envDefaults"]["attempts"] == 7 and o["envDefaults"]["backoffBaseMs"] == 250
    # base * 2^(attempt-1), +-30 % jitter, rounded; attempt 0 clamps to exponent 0
    assert o["backoffMid"] == [1000, 2000, 4000, 8000] and o["backoffLo"] == 700 and o["backoffHi"] == 1300 and o["backoffZero"] == 1000
    assert o["queueName"] == "image-restoration-jobs" and o["concurrency"] == 5
    # success: job record carries the reference's timing keys and the engine's request id; no pixels in the return value
    assert o["first"]["status"] == "succeeded" and set(o["first"]["timings"]) == {"classify_ms", "prompt_ms", "restore_ms", "total_ms"}
    last = o["firstLog"]["updates"][-1]
    assert last["status"] == "succeeded" and last["costUsd"] == 0 and last["signedResultUrl"].endswith("/j1") and last["prompt"] == "P"
    assert o["firstLog"]["calls"][0]["ctx"]["traceparent"] == "00-aa-bb-01" and o["firstLog"]["calls"][0]["prompt"] == "fix"
    # ONE object argument {imageBuffer, userPrompt, userContext, options} -- the reference's signature (restorator.js:37)
    assert o["firstLog"]["calls"][0]["options"] == {} and o["firstLog"]["calls"][0]["bytes"] == 3
    assert o["firstLog"]["stored"] == [6] and o["firstLog"]["dlq"] == [] and o["firstLog"]["refunds"] == []
    # retryable failures go back to queued, with the jittered-exponential delays, then succeed; gcs_ref goes through loadImage
    assert [u["status"] for u in o["retryLog"]["updates"]] == ["running", "queued", "running", "queued", "running", "succeeded"]
    assert o["retryDelays"] == [1000, 2000] and o["retryLog"]["calls"][0]["bytes"] == len("from:originals/u2/j2")
    # attempts exhausted: 4 delays, one DLQ entry, one refund, failed record with the error code
    assert o["exhausted"] is None and o["exhaustedDelays"] == [1000, 2000, 4000, 8000]
    dlq = o["exhaustedLog"]["dlq"]
    assert len(dlq) == 1 and dlq[0]["name"] == "failed-restoration" and dlq[0]["payload"]["originalJobId"] == "j3" and dlq[0]["payload"]["attempts"] == 5
    assert o["exhaustedLog"]["refunds"] == [["u3", "j3", 1]]
    assert o["exhaustedLog"]["updates"][-1]["status"] == "failed" and o["exhaustedLog"]["updates"][-1]["error"]["code"] == "ENGINE_X"
    # INVALID_INPUT and an empty payload are terminal on the first attempt
    assert o["invalidAttempts"] == 1 and [u["status"] for u in o["invalidLog"]["updates"]] == ["running", "failed"]
    assert o["invalidLog"]["refunds"] == [["u4", "j4", 1]] and len(o["invalidLog"]["dlq"]) == 1
    assert o["emptyAttempts"] == 1 and o["emptyLog"]["calls"] == [] and o["emptyLog"]["updates"][-1]["error"]["code"] == "INVALID_INPUT"
    assert "Worker class" in o["needsWorkerClass"]
