"""Host side of the preprocess step in front of the hot path -- the Python mirror of
server-node/src/middleware/imagePreprocess.js:24-91.

    decode (host codec) -> [GPU] EXIF auto-orient + fit inside 2048, Lanczos-3 -> JPEG q85 4:4:4 (host codec)

The pixel work runs in the engine (csrc/preprocess.hip through ire_preprocess); Pillow is only the codec here, as sharp
is in the reference.  The returned record mirrors what the middleware hangs on req.file (:70-78).
"""
import io

MAX_DIMENSION = 2048      # imagePreprocess.js:4
JPEG_QUALITY = 85         # imagePreprocess.js:5
EXIF_ORIENTATION = 0x0112


class PreprocessError(ValueError):
    """-> the middleware's 422 'Image Preprocessing Failed' problem (imagePreprocess.js:81-89); .status carries 400 for a
    missing file (:25-34)."""

    def __init__(self, message, status=422):
        super().__init__(message)
        self.status = status


def _js_round(x):
    import math
    return int(math.floor(x + 0.5))


def preprocess_image(engine, data, max_dim=MAX_DIMENSION, quality=JPEG_QUALITY):
    """encoded upload bytes -> dict(buffer=JPEG bytes, pixels=uint8 [H,W,3], operations=[...], original_metadata,
    processed_metadata, mimetype).  `engine` must be an image_restoration_platform_amd.engine.Engine (GPU)."""
    if not data:
        raise PreprocessError("An image file must be provided in the request.", status=400)
    import numpy as np
    from PIL import Image
    try:
        im = Image.open(io.BytesIO(data))
        im.load()
    except Exception as e:   # noqa: BLE001 -- any decoder failure is the same 422
        raise PreprocessError(str(e) or "Unable to preprocess the uploaded image.") from e
    fmt = (im.format or "").lower()
    orientation = 1
    try:
        orientation = int(im.getexif().get(EXIF_ORIENTATION, 1))
    except Exception:        # noqa: BLE001 -- unreadable EXIF: treat as upright, like failOnError:false (:40,:43)
        orientation = 1
    if orientation < 1 or orientation > 8:
        orientation = 1
    width, height = im.size                      # STORED dimensions, as sharp's metadata() (:40,:46)
    rgb = np.ascontiguousarray(np.asarray(im.convert("RGB")))
    operations = ["auto_orient"]
    out = engine.preprocess(rgb, orientation=orientation, max_dim=max_dim)
    if width > max_dim or height > max_dim:
        scale = max_dim / max(width, height)
        operations.append(f"resize_{_js_round(width * scale)}x{_js_round(height * scale)}")      # the box, as :54 logs it
    buf = io.BytesIO()
    # 4:4:4 = subsampling 0; mozjpeg's trellis/progressive tuning has no Pillow switch: optimize=True is the nearest setting
    # .withMetadata({icc: 'sRGB'}) (:65-68): the output carries an sRGB ICC profile (the standard one PIL / LittleCMS builds)
    from PIL import ImageCms
    icc = ImageCms.ImageCmsProfile(ImageCms.createProfile("sRGB")).tobytes()
    Image.fromarray(out).save(buf, format="JPEG", quality=quality, subsampling=0, optimize=True, icc_profile=icc)
    operations += [f"compress_jpeg_q{quality}", "attach_sRGB_icc"]
    return {
        "buffer": buf.getvalue(),
        "pixels": out,
        "operations": operations,
        "original_metadata": {"width": width, "height": height, "format": fmt, "orientation": orientation},
        "processed_metadata": {"width": int(out.shape[1]), "height": int(out.shape[0]), "format": "jpeg"},
        "mimetype": "image/jpeg",
    }
