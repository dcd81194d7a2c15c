"""CPU oracle for Fusion-v0 (numpy, integer arithmetic) -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED against the reference: the reference has no fusion code; its docs
only describe handing <= 3 images to the remote provider (image-restoration-platform.md:787-857,
geminiClient.js:32,49).  This is the independent restatement of the BUILD-DEFINED algorithm
documented at the top of image_restoration_platform_amd/csrc/fusion.hip; the HIP path must match it
bit for bit (shifts and pixels).
"""
import numpy as np

CR, FR, FM = 4, 3, 20


def luma(rgb):
    r, g, b = (rgb[..., c].astype(np.uint32) for c in range(3))
    return ((77 * r + 150 * g + 29 * b + 128) >> 8).astype(np.uint8)


def quarter(L):
    h, w = L.shape
    s = L.reshape(h // 4, 4, w // 4, 4).astype(np.uint32).sum(axis=(1, 3))
    return ((s + 8) >> 4).astype(np.uint8)


def _pick(sads, R):
    best = None
    for dy in range(-R, R + 1):
        for dx in range(-R, R + 1):
            key = (int(sads[dy + R, dx + R]), abs(dy) + abs(dx), dy, dx)
            if best is None or key < best:
                best = key
    return best[2], best[3]


def _sad_table(P0, Pv, R, M, step, by, bx):
    H, W = P0.shape
    ys = np.arange(M, H - M, step)
    xs = np.arange(M, W - M, step)
    a = P0[np.ix_(ys, xs)].astype(np.int64)
    out = np.zeros((2 * R + 1, 2 * R + 1), np.int64)
    for dy in range(-R, R + 1):
        for dx in range(-R, R + 1):
            yy = np.clip(ys + by + dy, 0, H - 1)
            xx = np.clip(xs + bx + dx, 0, W - 1)
            out[dy + R, dx + R] = np.abs(a - Pv[np.ix_(yy, xx)].astype(np.int64)).sum()
    return out


def wlut(noise):
    noise = min(max(float(noise), 0.0), 1.0) if noise == noise else 0.0
    sigma = 4.0 + 40.0 * noise
    d = np.arange(256, dtype=np.float64)
    w = np.floor(1024.0 * np.exp(-(d * d) / (2.0 * sigma * sigma)) + 0.5)
    return np.maximum(w, 1.0).astype(np.uint32)


def align(views):
    """-> shifts [k,2] (dy,dx): aligned sample of view v for ref pixel (y,x) is view_v[y+dy, x+dx]."""
    k = views.shape[0]
    L = [luma(v) for v in views]
    Q = [quarter(l) for l in L]
    shifts = np.zeros((k, 2), np.int32)
    for v in range(1, k):
        cdy, cdx = _pick(_sad_table(Q[0], Q[v], CR, CR, 1, 0, 0), CR)
        fdy, fdx = _pick(_sad_table(L[0], L[v], FR, FM, 2, 4 * cdy, 4 * cdx), FR)
        shifts[v] = (4 * cdy + fdy, 4 * cdx + fdx)
    return shifts


def fuse(views, noise_score):
    """views [k,H,W,3] uint8 (k = 2..3) -> (fused [H,W,3] uint8, shifts [k,2])."""
    views = np.asarray(views, dtype=np.uint8)
    k, H, W, _ = views.shape
    shifts = align(views)
    lut = wlut(noise_score).astype(np.int64)
    ys, xs = np.arange(H), np.arange(W)
    xsamp = []
    for v in range(k):
        yy = np.clip(ys + shifts[v, 0], 0, H - 1)
        xx = np.clip(xs + shifts[v, 1], 0, W - 1)
        xsamp.append(views[v][np.ix_(yy, xx)].astype(np.int64))
    if k == 2:
        m = (xsamp[0] + xsamp[1] + 1) >> 1
    else:
        a, b, d = xsamp
        m = np.maximum(np.minimum(a, b), np.minimum(np.maximum(a, b), d))
    num = np.zeros_like(m)
    den = np.zeros_like(m)
    for x in xsamp:
        w = lut[np.abs(x - m)]
        num += w * x
        den += w
    out = (num + (den >> 1)) // den
    return out.astype(np.uint8), shifts
