"""Synthetic workload generator: seeded 1/f-spectrum RGB fields with one
degradation per image index (none, box blur 5, +N(0,20) noise, x0.25 darken, channel gains).
There is no network for datasets; bench.py and the tests use these shapes and seeds."""
import numpy as np

BASE_SEED = 20251114


def pink_field(h, w, rng):
    """1/f-spectrum RGB field scaled to mean 110, sigma 45 (float64, not clipped)."""
    fy = np.fft.fftfreq(h)[:, None]
    fx = np.fft.rfftfreq(w)[None, :]
    f = np.sqrt(fy * fy + fx * fx)
    f[0, 0] = 1.0
    out = np.empty((h, w, 3), np.float64)
    for c in range(3):
        spec = (rng.standard_normal((h, w // 2 + 1)) + 1j * rng.standard_normal((h, w // 2 + 1))) / f
        spec[0, 0] = 0.0
        img = np.fft.irfft2(spec, s=(h, w))
        img = (img - img.mean()) / (img.std() + 1e-12)
        out[..., c] = 110.0 + 45.0 * img
    return out


def _box_blur5(img):
    p = np.pad(img, ((2, 2), (2, 2), (0, 0)), mode="edge")
    c = np.cumsum(np.cumsum(p, axis=0), axis=1)
    c = np.pad(c, ((1, 0), (1, 0), (0, 0)))
    h, w = img.shape[:2]
    return (c[5:5 + h, 5:5 + w] - c[0:h, 5:5 + w] - c[5:5 + h, 0:w] + c[0:h, 0:w]) / 25.0


def image(i, h, w, seed=BASE_SEED):
    rng = np.random.default_rng(seed + i)
    base = pink_field(h, w, rng)
    kind = i % 5
    if kind == 1:
        base = _box_blur5(base)
    elif kind == 2:
        base = base + rng.normal(0.0, 20.0, base.shape)
    elif kind == 3:
        base = base * 0.25
    elif kind == 4:
        base = base * np.array([1.3, 0.8, 0.6])[None, None, :]
    return np.clip(np.rint(base), 0, 255).astype(np.uint8)


def batch(n, h, w, seed=BASE_SEED, start=0):
    return np.stack([image(start + i, h, w, seed) for i in range(n)], axis=0)


def fusion_views(h, w, seed=BASE_SEED, shifts=((0, 0), (5, -3), (-4, 6)), noise_sigma=10.0):
    """3 views of one scene: integer shifts (dy,dx) and independent N(0,10) noise.
    view_v[y, x] = scene[y - dy_v, x - dx_v]  =>  aligned sample for ref pixel (y,x) is view_v[y+dy_v, x+dx_v]."""
    rng = np.random.default_rng(seed + 7777)
    m = 32
    scene = pink_field(h + 2 * m, w + 2 * m, rng)
    views = []
    for (dy, dx) in shifts:
        v = scene[m - dy:m - dy + h, m - dx:m - dx + w, :] + rng.normal(0.0, noise_sigma, (h, w, 3))
        views.append(np.clip(np.rint(v), 0, 255).astype(np.uint8))
    return np.stack(views, axis=0)
