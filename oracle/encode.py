"""CPU oracle of the device-side result encoder (image_restoration_platform_amd/csrc/encode.hip) -- TEST INFRASTRUCTURE, never on the
product path (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package).

The reference hands the restored image to its caller as a base64 STRING of an ENCODED image (server-node/src/services/restorator.js:108;
geminiClient.js:75-88 passes the provider's base64 through); which encoder is the provider's business, so there is nothing of the
reference's to restate byte for byte: parity with the reference is unpinned by nature.  What IS pinned, by published formats and their
stdlib implementations: the PNG container (ISO/IEC 15948: signature, IHDR, IDAT, IEND, CRC-32 per chunk), zlib (RFC 1950: header, Adler-32),
deflate STORED blocks (RFC 1951 3.2.4) and base64 (RFC 4648).  png_stored() builds the exact byte string the device writes from
zlib.crc32 / zlib.adler32 / struct; the tests also decode the device's output with PIL (an independent PNG reader) and compare pixels.
"""
import base64
import struct
import zlib

import numpy as np

STORED = 65535


def png_stored(rgb):
    """rgb uint8 [H, W, 3] -> bytes of a PNG file whose IDAT is ONE zlib stream of stored deflate blocks, scanline filter 0."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, c = rgb.shape
    assert c == 3
    raw = np.concatenate([np.zeros((h, 1), np.uint8), rgb.reshape(h, w * 3)], axis=1).tobytes()
    z = bytearray(b"\x78\x01")
    nblk = (len(raw) + STORED - 1) // STORED
    for b in range(nblk):
        part = raw[b * STORED:(b + 1) * STORED]
        z += struct.pack("<BHH", 1 if b + 1 == nblk else 0, len(part), len(part) ^ 0xFFFF) + part
    z += struct.pack(">I", zlib.adler32(raw) & 0xFFFFFFFF)

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", bytes(z)) + chunk(b"IEND", b"")


def png_base64(rgb):
    return base64.b64encode(png_stored(rgb))
