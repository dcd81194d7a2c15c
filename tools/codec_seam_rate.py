"""The deployed seam with a REAL codec: FastAPI's POST /restore end to end at 1024x1024 -- JPEG upload decoded by
PIL, classified and restored by the engine's batcher, the result encoded and base64'd as restorator.js:108 requires -- at 1 / 4 / 8
concurrent requests, per result codec (IRE_RESULT_CODEC): host PNG (PIL, zlib 6), host JPEG q85 4:4:4 (imagePreprocess.js:57-64), and
the device-side stored PNG + base64 (csrc/encode.hip).  Prints one JSON object; the codec's share of a request's wall time is measured
by timing the same decode / encode calls alone on this box's cores."""
import concurrent.futures
import io
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one_codec(codec, size, requests, levels):
    os.environ["IRE_RESULT_CODEC"] = codec
    import numpy as np
    from PIL import Image
    from fastapi.testclient import TestClient
    from image_restoration_platform_amd import restorator, synth
    from image_restoration_platform_amd.serving import app as appmod
    imgs = synth.batch(8, size, size)
    uploads = []
    for im in imgs:
        bio = io.BytesIO(); Image.fromarray(im).save(bio, format="JPEG", quality=85, subsampling=0)
        uploads.append(bio.getvalue())
    # the codec alone, one thread
    t0 = time.perf_counter()
    for u in uploads[:4]:
        restorator.decode_image(u)
    t_dec = (time.perf_counter() - t0) / 4
    t0 = time.perf_counter()
    n_enc = 2 if codec == "png" else 4
    for im in imgs[:n_enc]:
        (restorator.encode_jpeg_base64 if codec == "jpeg" else restorator.encode_png_base64)(im)
    t_enc_host = (time.perf_counter() - t0) / n_enc
    client = TestClient(appmod.app)
    r = client.post("/restore", content=uploads[0])          # engine start, first shapes
    assert r.status_code == 200, r.text[:300]
    out_chars = len(r.json()["restoredImage"])
    for u in uploads[1:4]:
        client.post("/restore", content=u)
    svc = appmod.get_service()
    rows = {}
    for conc in levels:
        t0 = time.perf_counter()
        with concurrent.futures.ThreadPoolExecutor(max_workers=conc) as ex:
            res = list(ex.map(lambda i: client.post("/restore", content=uploads[i % 8]).status_code, range(requests)))
        dt = time.perf_counter() - t0
        assert all(c == 200 for c in res)
        rows["http_%d_concurrent" % conc] = {"images_per_sec": requests / dt, "ms_per_request_wall": 1e3 * dt * conc / requests}
        # the same through the service layer (no HTTP framing / JSON of a 4-MB string)
        t0 = time.perf_counter()
        with concurrent.futures.ThreadPoolExecutor(max_workers=conc) as ex:
            ok = list(ex.map(lambda i: svc.restore(uploads[i % 8])["success"], range(requests)))
        dt = time.perf_counter() - t0
        assert all(ok)
        rows["service_%d_concurrent" % conc] = {"images_per_sec": requests / dt, "ms_per_request_wall": 1e3 * dt * conc / requests}
    t_enc = t_enc_host
    if codec == "png-device":
        eng = appmod._state["engine"]
        t0 = time.perf_counter()
        for im in imgs[:4]:
            eng.encode_png_base64(im)
        t_enc = (time.perf_counter() - t0) / 4               # host pixels -> device -> text -> host (the flagged batcher skips even that)
    return {"codec": codec, "result_chars": out_chars, "decode_jpeg_ms": 1e3 * t_dec, "encode_ms": 1e3 * t_enc,
            "encode_host_equivalent_ms": 1e3 * t_enc_host, "rates": rows, "cores": len(os.sched_getaffinity(0))}


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    if len(sys.argv) > 2:            # child: one codec per process (the codec is read at import, the engine is created once)
        print(json.dumps(one_codec(sys.argv[2], size, int(sys.argv[3]), [int(x) for x in sys.argv[4].split(",")])))
        return
    out = {}
    for codec, requests in (("png", 16), ("jpeg", 48), ("png-device", 48)):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), str(size), codec, str(requests), "1,4,8"], capture_output=True, text=True, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        out[codec] = json.loads(line[-1]) if line else {"error": (r.stderr or r.stdout)[-600:]}
    print(json.dumps({"workload": "FastAPI POST /restore @%dx%d, JPEG q85 upload, 1 GPU" % (size, size), "by_result_codec": out}))


if __name__ == "__main__":
    main()
