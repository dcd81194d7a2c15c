"""CPU oracle (TEST INFRASTRUCTURE ONLY -- never imported by the product) of the preprocess step that feeds the hot
path: server-node/src/middleware/imagePreprocess.js:24-91 = auto-orient, fit inside 2048 with a Lanczos-3 kernel,
JPEG q85 4:4:4.  This file restates the two PIXEL operations (orientation, resample) and the size rule.

Pinning.  The reference delegates the pixel work to sharp/libvips (package.json:37), which is not in the tree and not
installable here, so parity with libvips' own reducer is UNPINNED.  What this oracle is pinned against instead
(tests/test_preprocess.py): Pillow 12's `Image.resize(..., LANCZOS)` and `Image.transpose` -- bit-exact -- because Pillow
is importable in this image and its 8-bit resampler is a published, integer-coefficient algorithm (two passes,
horizontal then vertical, 22-bit fixed-point taps, one rounding to u8 per pass).  The size rule follows the
reference's own arithmetic (imagePreprocess.js:12-22,46-55) including its use of PRE-rotation metadata for the box.
"""
import math

import numpy as np

MAX_DIMENSION = 2048          # imagePreprocess.js:4
JPEG_QUALITY = 85             # imagePreprocess.js:5
PRECISION_BITS = 32 - 8 - 2   # Pillow: 8-bit samples, 2 guard bits
LANCZOS_SUPPORT = 3.0


def js_round(x):
    """Math.round: half towards +inf."""
    return int(math.floor(x + 0.5))


def plan(width, height, orientation=1, max_dim=MAX_DIMENSION):
    """-> (out_w, out_h, resized).  (width, height) are the STORED dimensions (sharp metadata() does not apply
    EXIF orientation); the box is computed from them (imagePreprocess.js:12-22) and the oriented image is then fitted
    inside that box without enlargement (:48-53)."""
    ow, oh = (height, width) if orientation in (5, 6, 7, 8) else (width, height)
    if not width or not height or (width <= max_dim and height <= max_dim):
        return ow, oh, False
    scale = max_dim / max(width, height)
    bw, bh = js_round(width * scale), js_round(height * scale)
    s2 = min(bw / ow, bh / oh, 1.0)
    return max(1, js_round(ow * s2)), max(1, js_round(oh * s2)), True


def orient(rgb, orientation):
    """EXIF orientation 1..8 -> upright pixels (what sharp's .rotate() with no angle does, imagePreprocess.js:43)."""
    a = np.asarray(rgb)
    if orientation == 2:
        a = a[:, ::-1]
    elif orientation == 3:
        a = a[::-1, ::-1]
    elif orientation == 4:
        a = a[::-1]
    elif orientation == 5:
        a = a.transpose(1, 0, 2)
    elif orientation == 6:
        a = a.transpose(1, 0, 2)[:, ::-1]       # rotate 90 deg clockwise
    elif orientation == 7:
        a = a.transpose(1, 0, 2)[::-1, ::-1]
    elif orientation == 8:
        a = a.transpose(1, 0, 2)[::-1]          # rotate 90 deg counter-clockwise
    return np.ascontiguousarray(a)


def _lanczos(x):
    if -LANCZOS_SUPPORT <= x < LANCZOS_SUPPORT:
        if x == 0.0:
            return 1.0
        px = x * math.pi
        a = math.sin(px) / px
        py = px / LANCZOS_SUPPORT
        return a * (math.sin(py) / py)
    return 0.0


def coefficients(in_size, out_size):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc -> (ksize, bounds [out,2] int32 (first, count), taps [out,ksize] int32)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = LANCZOS_SUPPORT * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    taps = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_lanczos((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x, v in enumerate(w):
            k = v / ww if ww != 0.0 else v
            taps[xx, x] = int(k * (1 << PRECISION_BITS) - 0.5) if k < 0 else int(k * (1 << PRECISION_BITS) + 0.5)
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, taps


def _pass(img, out_size, axis):
    in_size = img.shape[axis]
    _, bounds, taps = coefficients(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for xx in range(out_size):
        x0, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = np.tensordot(taps[xx, :n].astype(np.int64), src[x0:x0 + n], axes=(0, 0)) + (1 << (PRECISION_BITS - 1))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize(rgb, out_w, out_h):
    """Two-pass integer Lanczos-3: horizontal, then vertical (each pass only if that size changes, as Pillow)."""
    a = np.ascontiguousarray(rgb)
    if a.shape[1] != out_w:
        a = _pass(a, out_w, 1)
    if a.shape[0] != out_h:
        a = _pass(a, out_h, 0)
    return np.ascontiguousarray(a)


def preprocess_pixels(rgb, orientation=1, max_dim=MAX_DIMENSION):
    """stored pixels [H,W,3] u8 -> (upright, fitted pixels, operations list as imagePreprocess.js:41-66 names them)."""
    h, w, _ = rgb.shape
    ow, oh, resized = plan(w, h, orientation, max_dim)
    out = orient(rgb, orientation)
    ops = ["auto_orient"]
    if resized:
        out = resize(out, ow, oh)
        scale = max_dim / max(w, h)
        ops.append(f"resize_{js_round(w * scale)}x{js_round(h * scale)}")
    ops += [f"compress_jpeg_q{JPEG_QUALITY}", "attach_sRGB_icc"]
    return out, ops
