"""FastAPI host of the engine -- the server-python path of BASELINE.json's north_star.

Extends the reference's stub (server-python/main.py:1-7: GET /health -> {"ok": true, "service": "python"})
with the endpoints its docs sketch (image-restoration-platform.md:1076-1127):
  POST /restore   body = one encoded image (JPEG/PNG/WebP bytes), optional ?prompt=  -> RestoratorService envelope
                  ?preprocess=1 first runs the upload through the preprocess step of imagePreprocess.js:24-91
                  (auto-orient + fit inside 2048 on the GPU, JPEG q85 4:4:4) as the Node middleware does before queueing
  POST /fuse      JSON {"images": [base64, ...2..3], "prompt": "..."}                 -> {base64Image, metadata}
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
_state = {"engine": None, "service": None, "error": None}


def get_service():
    with _lock:
        if _state["service"] is None:
            try:
                eng = Engine(device_index=0, max_batch=8)
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
