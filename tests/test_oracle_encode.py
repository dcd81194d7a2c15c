"""oracle/encode.py (the checker of csrc/encode.hip) against the formats it restates: an independent PNG reader (PIL) must decode its
file to the same pixels, and its framing must be what RFC 1950 / 1951 / ISO 15948 prescribe (one IDAT, stored blocks, both checksums)."""
import base64
import io
import struct
import zlib

import numpy as np
import pytest

from oracle import encode as oenc


@pytest.mark.parametrize("h,w", [(1, 8), (17, 8), (64, 96), (21, 7280), (300, 512)])
def test_oracle_png_is_a_png(h, w):
    from PIL import Image
    img = np.random.default_rng(h + w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    f = oenc.png_stored(img)
    assert np.array_equal(np.asarray(Image.open(io.BytesIO(f)).convert("RGB")), img)
    assert f[:8] == b"\x89PNG\r\n\x1a\n" and f[12:16] == b"IHDR" and f[37:41] == b"IDAT" and f[-12:] == b"\x00\x00\x00\x00IEND\xaeB`\x82"
    zlen = struct.unpack(">I", f[33:37])[0]
    z = f[41:41 + zlen]
    raw = zlib.decompress(z)                                  # zlib itself accepts the stored stream (Adler-32 checked by it)
    assert len(raw) == h * (1 + 3 * w) and raw[0] == 0
    assert struct.unpack(">I", f[41 + zlen:45 + zlen])[0] == zlib.crc32(f[37:41 + zlen]) & 0xFFFFFFFF
    assert base64.b64decode(oenc.png_base64(img)) == f
    assert len(f) == 57 + 2 + 5 * ((len(raw) + 65534) // 65535) + len(raw) + 4
