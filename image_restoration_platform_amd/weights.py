"""RestoreNet-v0: architecture spec, seeded weights and the engine's weight-file format.

BUILD-DEFINED: the reference's restoration step is a remote call to
`gemini-2.5-flash-image` (server-node/src/clients/geminiClient.js:32-97) and has no local
network or weights, so this module *defines* the network that stands behind the same seam.
Restoration-pixel parity is therefore UNPINNED against the reference; the CPU oracle
(oracle/restorenet.py) and the HIP engine are checked against each other on these weights.

Architecture (NHWC, bf16 storage, fp32 accumulate; widths 32/64/128/256 at scales 1,1/2,1/4,1/8):
  stem   conv3x3 3->32 on the raw 0..255 pixel values (zero padding, 1/255 folded into the weights)
  enc l  2 x ResBlock(C_l), l = 0..3;  down l: conv3x3 stride 2 C_l -> C_{l+1}, l = 0..2
  mid    2 x ResBlock(256)
  dec l  (l = 2,1,0): up l = nearest x2 -> conv3x3 C_{l+1} -> C_l; fuse l = conv1x1 on
         concat(up, skip_l) 2C_l -> C_l; 2 x ResBlock(C_l)
  head   GroupNorm -> SiLU -> conv3x3 32 -> 3 ; out = clamp(round(input + head), 0, 255)
  ResBlock(C): x + conv2(silu(gn2(conv1(silu(gn1(x))))));  GroupNorm: 8 groups, eps 1e-5,
         affine, then FiLM: y*(1+s_l) + t_l with (s_l, t_l) = slices of Linear(7->960)(scores)
43 convolutions, 33 GroupNorms, 779.9 GFLOP per 1024x1024 image.

All tensors are float32 in the file; conv weights are rounded to bf16-representable values at
generation so the oracle and the engine multiply identical numbers.
"""
import os
import struct

import numpy as np

WIDTHS = (32, 64, 128, 256)
NUM_GROUPS = 8
GN_EPS = 1e-5
FILM_DIM = 2 * sum(WIDTHS)  # 960
FILM_OFFSETS = (0, 64, 192, 448)  # level l: scale at off..off+C, shift at off+C..off+2C
MAGIC = b"IREW"
VERSION = 1


def resblock_names(prefix):
    return [f"{prefix}.gn1.g", f"{prefix}.gn1.b", f"{prefix}.conv1.w", f"{prefix}.conv1.b",
            f"{prefix}.gn2.g", f"{prefix}.gn2.b", f"{prefix}.conv2.w", f"{prefix}.conv2.b"]


def spec():
    """Ordered list of (name, shape, kind)."""
    out = [("stem.w", (32, 3, 3, 3), "stem_w"), ("stem.b", (32,), "stem_b")]

    def rb(prefix, c):
        return [(f"{prefix}.gn1.g", (c,), "gamma"), (f"{prefix}.gn1.b", (c,), "beta"),
                (f"{prefix}.conv1.w", (c, c, 3, 3), "conv_w"), (f"{prefix}.conv1.b", (c,), "bias"),
                (f"{prefix}.gn2.g", (c,), "gamma"), (f"{prefix}.gn2.b", (c,), "beta"),
                (f"{prefix}.conv2.w", (c, c, 3, 3), "conv_w"), (f"{prefix}.conv2.b", (c,), "bias")]

    for l, c in enumerate(WIDTHS):
        for i in range(2):
            out += rb(f"enc{l}.rb{i}", c)
        if l < 3:
            out += [(f"down{l}.w", (WIDTHS[l + 1], c, 3, 3), "conv_w"), (f"down{l}.b", (WIDTHS[l + 1],), "bias")]
    for i in range(2):
        out += rb(f"mid.rb{i}", 256)
    for l in (2, 1, 0):
        c = WIDTHS[l]
        out += [(f"up{l}.w", (c, WIDTHS[l + 1], 3, 3), "conv_w"), (f"up{l}.b", (c,), "bias"),
                (f"fuse{l}.w", (c, 2 * c, 1, 1), "conv_w"), (f"fuse{l}.b", (c,), "bias")]
        for i in range(2):
            out += rb(f"dec{l}.rb{i}", c)
    out += [("head.gn.g", (32,), "gamma"), ("head.gn.b", (32,), "beta"),
            ("head.w", (3, 32, 3, 3), "head_w"), ("head.b", (3,), "head_b"),
            ("film.w", (FILM_DIM, 7), "film_w"), ("film.b", (FILM_DIM,), "film_b")]
    return out


def round_to_bf16(a):
    """float32 array -> float32 array of bf16-representable values (round to nearest even)."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(np.shape(a))


def generate(seed=0):
    """Deterministic seeded init (numpy PCG64; no torch RNG so the bytes never change)."""
    rng = np.random.default_rng(seed)
    w = {}
    for name, shape, kind in spec():
        if kind == "conv_w":
            fan_in = shape[1] * shape[2] * shape[3]
            t = rng.standard_normal(shape) * (0.5 * np.sqrt(2.0 / fan_in))
            t = round_to_bf16(t.astype(np.float32))
        elif kind == "stem_w":  # acts on raw 0..255 values
            t = rng.standard_normal(shape) * (0.5 * np.sqrt(2.0 / 27.0) / 64.0)
            t = round_to_bf16(t.astype(np.float32))
        elif kind == "head_w":  # output is in 0..255 units; std ~10 levels on unit-variance features
            t = rng.standard_normal(shape) * 1.0
            t = round_to_bf16(t.astype(np.float32))
        elif kind in ("bias", "stem_b", "head_b"):
            t = rng.standard_normal(shape) * (0.5 if kind == "stem_b" else 0.02)
        elif kind == "gamma":
            t = 1.0 + 0.1 * rng.standard_normal(shape)
        elif kind == "beta":
            t = 0.1 * rng.standard_normal(shape)
        elif kind == "film_w":
            t = 0.1 * rng.standard_normal(shape)
        elif kind == "film_b":
            t = np.zeros(shape)
        else:
            raise AssertionError(kind)
        w[name] = np.ascontiguousarray(t, dtype=np.float32)
    return w


def serialize(weights):
    parts = [MAGIC, struct.pack("<II", VERSION, len(weights))]
    for name, shape, _ in spec():
        a = np.ascontiguousarray(weights[name], dtype="<f4")
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"{name}: shape {a.shape} != {shape}")
        nb = name.encode()
        nb += b"\0" * ((-len(nb)) % 4)
        parts.append(struct.pack("<I", len(nb)) + nb)
        parts.append(struct.pack("<I", a.ndim) + struct.pack(f"<{a.ndim}I", *a.shape))
        parts.append(a.tobytes())
    return b"".join(parts)


def deserialize(blob):
    if blob[:4] != MAGIC:
        raise ValueError("invalid weight file: bad magic")
    ver, n = struct.unpack_from("<II", blob, 4)
    if ver != VERSION:
        raise ValueError("invalid weight file: version")
    off = 12
    out = {}
    for _ in range(n):
        (ln,) = struct.unpack_from("<I", blob, off); off += 4
        name = blob[off:off + ln].rstrip(b"\0").decode(); off += ln
        (nd,) = struct.unpack_from("<I", blob, off); off += 4
        dims = struct.unpack_from(f"<{nd}I", blob, off); off += 4 * nd
        cnt = int(np.prod(dims))
        out[name] = np.frombuffer(blob, dtype="<f4", count=cnt, offset=off).reshape(dims).copy(); off += 4 * cnt
    return out


def default_path(seed=0):
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "weights")
    return os.path.join(d, f"restorenet_v0_seed{seed}.bin")


def ensure_default(seed=0):
    """Write the seeded weight file if it is not there yet; return its path."""
    p = default_path(seed)
    if not os.path.exists(p):
        os.makedirs(os.path.dirname(p), exist_ok=True)
        tmp = p + f".tmp{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(serialize(generate(seed)))
        os.replace(tmp, p)
    return p


def conv_flops(h, w):
    """(flops_3x3, flops_1x1) per image for an h x w input (BASELINE.md section 3 formulas)."""
    f3 = f1 = 0.0
    for l, c in enumerate(WIDTHS):
        px = (h >> l) * (w >> l)
        n_cc = 8 if l == 3 else 8  # enc 4 + (mid 4 | dec 4) C->C convs per level
        f3 += n_cc * 2 * 9 * c * c * px
    f3 += 2 * (2 * 9 * 3 * 32 * h * w)  # stem + head
    for l in range(3):
        px_lo = (h >> (l + 1)) * (w >> (l + 1))
        px_hi = (h >> l) * (w >> l)
        f3 += 2 * 9 * WIDTHS[l] * WIDTHS[l + 1] * px_lo  # down
        f3 += 2 * 9 * WIDTHS[l + 1] * WIDTHS[l] * px_hi  # up
        f1 += 2 * 2 * WIDTHS[l] * WIDTHS[l] * px_hi      # fuse
    return f3, f1
