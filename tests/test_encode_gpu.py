"""The device-side result encoder (csrc/encode.hip): restored pixels -> the base64 text of a PNG file (stored deflate blocks),
bit-exact against oracle/encode.py (struct + zlib.crc32 + zlib.adler32 + base64: the published formats' stdlib implementations)
and decoded back by PIL, an independent PNG reader.  The reference's result contract: restorator.js:108 (`restoredImage` is a base64
string of an encoded image; geminiClient.js:75-88); which encoder is the provider's business => parity with the reference unpinned."""
import base64
import io

import numpy as np
import pytest

from image_restoration_platform_amd import _lib, synth
from oracle import encode as oenc

pytestmark = pytest.mark.gpu


def _check(text, img):
    ref = oenc.png_base64(img)
    assert len(text) == len(ref)
    assert text == ref, next(i for i in range(len(ref)) if text[i] != ref[i])
    from PIL import Image
    back = np.asarray(Image.open(io.BytesIO(base64.b64decode(text))).convert("RGB"))
    assert np.array_equal(back, img)


@pytest.mark.parametrize("h,w", [(16, 16), (64, 64), (17, 8), (1, 8), (203, 104), (256, 256), (21, 7280), (22, 7280), (1024, 1024)])
def test_png_base64_is_bit_exact(engine, h, w):
    """Sizes around every boundary of the format: raw streams of 1 .. 49 stored blocks, a block boundary in the middle of a scanline,
    a last block of 1 byte short of full (21 x 7280: 21 x 21841 = 458 661 = 7 x 65535 - 84), files whose length is 0 / 1 / 2 mod 3 (the
    base64 tail), CRC ranges of one slice, one workgroup and many workgroups with a ragged tail."""
    rng = np.random.default_rng(h * 10007 + w)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    assert engine.png_base64_bytes(h, w) == len(oenc.png_base64(img))
    _check(engine.encode_png_base64(img), img)


def test_png_base64_extreme_pixels_and_batches(engine):
    for fill in (0, 255):
        img = np.full((40, 64, 3), fill, np.uint8)
        _check(engine.encode_png_base64(img), img)
    imgs = synth.batch(3, 72, 136, start=2)
    texts = engine.encode_png_base64(imgs)
    for t, im in zip(texts, imgs):
        _check(t, im)
    # twice in a row: the checksum state of the first call is gone (tickets and sums start from zero again)
    texts2 = engine.encode_png_base64(imgs[::-1].copy())
    for t, im in zip(texts2, imgs[::-1]):
        _check(t, im)


def test_png_base64_device_entry_and_errors(engine):
    import torch
    from image_restoration_platform_amd.engine import EngineError
    imgs = synth.batch(2, 64, 96, start=9)
    out = engine.encode_png_base64_tensor(torch.from_numpy(imgs).cuda())
    torch.cuda.synchronize()
    for i in range(2):
        _check(out[i].cpu().numpy().tobytes(), imgs[i])
    assert engine.png_base64_bytes(64, 60) == 0                       # width not a multiple of 8
    with pytest.raises(EngineError) as e:
        engine.encode_png_base64(np.zeros((16, 12, 3), np.uint8))
    assert e.value.status == 1 and "invalid" in e.value.message


def test_batcher_delivers_png_text_when_the_engine_is_flagged():
    """IRE_FLAG_RESULT_PNG_BASE64: ire_submit / ire_poll hand back the string restorator.js:108 puts on the wire -- the base64 text of a
    PNG of the restored image -- encoded on the device behind the restore; it decodes to exactly the pixels an unflagged engine returns."""
    from image_restoration_platform_amd.engine import Engine
    imgs = synth.batch(5, 64, 96, start=40)
    plain = Engine(max_batch=4)
    try:
        sc, _ = plain.classify(imgs[:4], True)
        ref = plain.restore(imgs[:4], scores=sc)
    finally:
        plain.close()
    eng = Engine(max_batch=4, flags=_lib.IRE_FLAG_RESULT_PNG_BASE64)
    try:
        jobs = [eng.submit(imgs[i], scores=sc[i]) for i in range(4)]
        for i, job in enumerate(jobs):
            text, scores, _ = eng.poll(job, timeout_ms=60000)
            _check(text, ref[i])
            assert np.array_equal(scores, sc[i])
        # the synchronous entry points are unaffected by the flag
        assert np.array_equal(eng.restore(imgs[:2], scores=sc[:2]), ref[:2])
    finally:
        eng.close()
