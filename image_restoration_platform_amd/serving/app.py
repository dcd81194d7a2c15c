"""FastAPI host of the engine -- the server-python path of BASELINE.json's north_star.

Extends the reference's stub (server-python/main.py:1-7: GET /health -> {"ok": true, "service": "python"})
with the endpoints its docs sketch (image-restoration-platform.md:1076-1127):
  POST /restore   body = one encoded image (JPEG/PNG/WebP bytes), optional ?prompt=  -> RestoratorService envelope
                  ?preprocess=1 first runs the upload through the preprocess step of imagePreprocess.js:24-91
                  (auto-orient + fit inside 2048 on the GPU, JPEG q85 4:4:4) as the Node middleware does before queueing
  POST /fuse      JSON {"images": [base64, ...2..3], "prompt": "..."}                 -> {base64Image, metadata}
  POST /restore_batch  JSON {"images": [base64, ...], "prompt": "..."}                -> [RestoratorService envelope, ...] in order
                  restoreBatch (restorator.js:181-236) through the PyTorch-ROCm extension (torch_host.TorchEngine): images of
                  one shape are stacked into ONE uint8 tensor, classified and restored as one engine batch on the GPU, and
                  (widths that are multiples of 8, no crop) their base64 PNG texts are written on the device too
  GET  /health/ready                                                                 -> service + engine health
The engine is created on first use; without a gfx950 device every compute endpoint answers 503 with the
engine's "service unavailable" message (there is no CPU fallback).
Run: uvicorn image_restoration_platform_amd.serving.app:app
"""
import base64
import threading

from fastapi import FastAPI, Request
from fastapi.responses import JSONResponse

from ..engine import Engine, EngineError
from ..restorator import EngineRestorer, RestoratorService

app = FastAPI(title="image-restoration engine (MI355X)")
_lock = threading.Lock()
_state = {"engine": None, "service": None, "error": None, "torch_engine": None}


def get_torch_engine():
    """The tensor host (csrc/torch_ext.cpp): created on first use, beside the ctypes engine the single-image endpoints use."""
    with _lock:
        if _state["torch_engine"] is None:
            from ..torch_host import TorchEngine
            _state["torch_engine"] = TorchEngine(device_index=0, max_batch=8)
        return _state["torch_engine"]


def get_service():
    with _lock:
        if _state["service"] is None:
            try:
                from ..restorator import RESULT_CODEC
                # IRE_RESULT_CODEC=png-device: the batcher returns the base64 text of a device-encoded PNG (csrc/encode.hip)
                eng = Engine(device_index=0, max_batch=8, flags=1 if RESULT_CODEC == "png-device" else 0)
            except Exception as e:  # noqa: BLE001 -- EngineError or a missing library
                _state["error"] = str(e)
                raise
            _state["engine"] = eng
            _state["service"] = RestoratorService(engine=eng)
        return _state["service"]


def _problem(status, title, detail):
    # RFC 7807 shape, like the reference's utils/problem.js
    return JSONResponse(status_code=status, content={"type": "about:blank", "title": title, "status": status, "detail": detail},
                        media_type="application/problem+json")


@app.get("/health")
def health():
    return {"ok": True, "service": "python"}


@app.get("/health/ready")
def ready():
    try:
        svc = get_service()
    except Exception as e:  # noqa: BLE001
        return JSONResponse(status_code=503, content={"ok": False, "engine": False, "detail": str(e)})
    st = svc.get_health_status()
    ok = bool(st["services"].get("engine"))
    # the shape of the Node side's GET /health/ready (healthRouter.js:80-117): status + dependencies + metrics
    dep = {"status": "ok" if ok else "unavailable", "device": "gfx950", **(st.get("metrics", {}).get("engine") or {})}
    body = {"ok": ok, "status": "ok" if ok else "unready", "dependencies": {"engine": dep}, **st}
    return body if ok else JSONResponse(status_code=503, content=body)


@app.post("/restore")
async def restore(request: Request, prompt: str = None, preprocess: int = 0):
    body = await request.body()
    if not body:
        return _problem(400, "Bad Request", "invalid request: empty body, expected encoded image bytes")
    try:
        svc = get_service()
    except Exception as e:  # noqa: BLE001
        return _problem(503, "Service Unavailable", str(e))
    operations = None
    if preprocess:
        from ..preprocess import PreprocessError, preprocess_image
        try:
            rec = preprocess_image(_state["engine"], body)
        except PreprocessError as e:
            return _problem(e.status, "Image Preprocessing Failed", str(e))
        except EngineError as e:
            return _problem(503 if e.status == 3 else 400 if e.status == 1 else 500, "Engine error", e.message)
        body, operations = rec["buffer"], rec["operations"]
    result = svc.restore(body, user_prompt=prompt, user_context={"userId": request.headers.get("x-user-id")})
    if not result["success"]:
        code = {"INVALID_INPUT": 400, "SERVICE_UNAVAILABLE": 503, "TIMEOUT": 504}.get(result["error"]["type"], 500)
        return JSONResponse(status_code=code, content=result)
    if operations is not None:
        result["metadata"]["preprocessOperations"] = operations        # req.file.preprocessOperations, imagePreprocess.js:78
    return result


@app.post("/restore_batch")
async def restore_batch(request: Request):
    """restoreBatch (restorator.js:181-236): same envelope per image, same order; the images go to the GPU as tensors."""
    import time
    import numpy as np
    import torch
    from ..prompt_enhancer import KEYS, PromptEnhancerService
    from ..restorator import decode_image, encode_png_base64, pad_to_multiple
    try:
        payload = await request.json()
        bufs = [base64.b64decode(b) for b in payload["images"]]
        if not bufs:
            raise ValueError("empty")
    except Exception:  # noqa: BLE001
        return _problem(400, "Bad Request", "invalid request: expected JSON {images: [base64, ...]}")
    try:
        te = get_torch_engine()
    except Exception as e:  # noqa: BLE001
        return _problem(503, "Service Unavailable", str(e))
    enhancer = PromptEnhancerService(None)
    results = [None] * len(bufs)
    decoded, groups = {}, {}
    for i, b in enumerate(bufs):
        try:
            rgb, fmt = decode_image(b)
            padded, hw = pad_to_multiple(rgb)
            decoded[i] = (padded, hw, fmt, rgb)
            groups.setdefault(padded.shape, []).append(i)
        except Exception as e:  # noqa: BLE001 -- the reference's envelope for a failed image (restorator.js:141-167)
            results[i] = {"success": False, "error": {"message": str(e), "code": "RESTORATION_FAILED", "type": "INVALID_INPUT"},
                          "timings": {"total_ms": 0}, "metadata": {"processingTime": 0, "failureStage": "CLASSIFICATION"}}
    for shape, idxs in groups.items():
        for c0 in range(0, len(idxs), 8):
            chunk = idxs[c0:c0 + 8]
            t0 = time.time()
            try:
                x = torch.from_numpy(np.stack([decoded[i][0] for i in chunk])).cuda()
                jp = torch.tensor([1 if decoded[i][2] == "jpeg" else 0 for i in chunk], dtype=torch.uint8, device="cuda")
                # condition on each image's own scores (unpadded pixels), as analyze() reports them
                same = all(decoded[i][0].shape == decoded[i][3].shape for i in chunk)
                if same:
                    scores, _ = te.classify(x, jp)
                else:
                    scores = torch.cat([te.classify(torch.from_numpy(decoded[i][3][None]).cuda(), jp[k:k + 1])[0] for k, i in enumerate(chunk)])
                t1 = time.time()
                restored = te.restore(x, scores, None)
                # the result text on the device when the chunk's images need no crop (csrc/encode.hip through the extension):
                # the D2H copy carries base64 characters, the host encodes nothing
                texts = None
                if same and restored.shape[2] % 8 == 0:
                    texts = te.encode_png_base64(restored).cpu().numpy()
                out = restored.cpu().numpy() if texts is None else None
                sc = scores.cpu().numpy()
                t2 = time.time()
            except EngineError as e:
                for i in chunk:
                    results[i] = {"success": False, "error": {"message": e.message, "code": e.code, "type": RestoratorService._classify_error(e)},
                                  "timings": {}, "metadata": {"processingTime": 0, "failureStage": "AI_RESTORATION"}}
                continue
            for k, i in enumerate(chunk):
                h, w = decoded[i][1]
                degradation = {key: float(sc[k, j]) for j, key in enumerate(KEYS)}
                results[i] = {
                    "success": True, "restoredImage": (texts[k].tobytes().decode("ascii") if texts is not None
                                                       else encode_png_base64(np.ascontiguousarray(out[k, :h, :w]))),
                    "degradationAnalysis": degradation,
                    "enhancedPrompt": enhancer.enhance(degradation=degradation, user_prompt=payload.get("prompt"), options={"batchIndex": i, "batchSize": len(bufs)}),
                    "timings": {"classify_ms": int(1e3 * (t1 - t0)), "prompt_ms": 0, "restore_ms": int(1e3 * (t2 - t1)), "total_ms": int(1e3 * (t2 - t0))},
                    "metadata": {"providerRequestId": f"ire-batch-{i}", "estimatedCostUsd": 0, "billedTokens": None, "processingTime": int(1e3 * (t2 - t0)),
                                 "classificationIssues": [{"type": key, "confidence": v} for key, v in degradation.items() if v > 0.3]}}
    return results


@app.post("/fuse")
async def fuse(request: Request):
    try:
        payload = await request.json()
        images = [base64.b64decode(b) for b in payload["images"]]
    except Exception:  # noqa: BLE001
        return _problem(400, "Bad Request", "invalid request: expected JSON {images: [base64 x 2..3]}")
    try:
        svc = get_service()
        out = EngineRestorer(_state["engine"]).restore_image(payload.get("prompt", ""), images, None)
        del svc
    except EngineError as e:
        return _problem(503 if e.status == 3 else 400 if e.status == 1 else 500, "Engine error", e.message)
    except ValueError as e:
        return _problem(400, "Bad Request", str(e))
    except Exception as e:  # noqa: BLE001
        return _problem(503, "Service Unavailable", str(e))
    return out
