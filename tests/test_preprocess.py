"""Preprocess step in front of the hot path (imagePreprocess.js:24-91).
CPU: the oracle is pinned bit-exactly against Pillow (the reference's sharp/libvips is absent: parity with it unpinned).
GPU: the engine (csrc/preprocess.hip through the C ABI) is bit-exact against the oracle."""
import io

import numpy as np
import pytest

from oracle import preprocess as opp


def _img(h, w, seed=0):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    base[..., 0] = (base[..., 0] // 4 + (xx * 255 // max(w - 1, 1)) * 3 // 4).astype(np.uint8)   # structure + noise
    base[..., 1] = (base[..., 1] // 2 + (yy * 255 // max(h - 1, 1)) // 2).astype(np.uint8)
    return base


@pytest.mark.parametrize("h,w,oh,ow", [(97, 131, 40, 53), (64, 64, 64, 31), (300, 200, 77, 200), (50, 70, 50, 70), (33, 47, 11, 9),
                                        (10, 400, 3, 57), (257, 13, 100, 13)])
def test_oracle_resize_matches_pillow_bit_exact(h, w, oh, ow):
    from PIL import Image
    a = _img(h, w, seed=h * 1000 + w)
    ref = np.asarray(Image.fromarray(a).resize((ow, oh), Image.LANCZOS))
    assert np.array_equal(opp.resize(a, ow, oh), ref)


def test_oracle_orientation_matches_pillow():
    from PIL import Image
    t = {2: Image.FLIP_LEFT_RIGHT, 3: Image.ROTATE_180, 4: Image.FLIP_TOP_BOTTOM, 5: Image.TRANSPOSE, 6: Image.ROTATE_270,
         7: Image.TRANSVERSE, 8: Image.ROTATE_90}         # the table of PIL.ImageOps.exif_transpose
    a = _img(5, 7)
    assert np.array_equal(opp.orient(a, 1), a)
    for o, m in t.items():
        assert np.array_equal(opp.orient(a, o), np.asarray(Image.fromarray(a).transpose(m))), o


def test_plan_follows_the_reference_arithmetic():
    # imagePreprocess.js:12-22: scale = 2048 / max(w, h), Math.round both sides; no resize at or below 2048
    assert opp.plan(4000, 3000) == (2048, 1536, True)
    assert opp.plan(2048, 2048) == (2048, 2048, False)
    assert opp.plan(2049, 100) == (2048, 100, True)
    assert opp.plan(3001, 5000) == (1229, 2048, True)       # 3001 * 0.4096 = 1229.2
    assert opp.plan(1000, 800, orientation=6) == (800, 1000, False)
    # the box is computed from the STORED size (:46-47) but applied after .rotate() (:43,:48-53): a rotated 4000x3000
    # upload must fit inside 2048x1536, i.e. comes out 1152x1536
    assert opp.plan(4000, 3000, orientation=6) == (1152, 1536, True)
    assert opp.plan(4000, 3000, orientation=3) == (2048, 1536, True)


def test_abi_plan_matches_oracle_without_gpu():
    from image_restoration_platform_amd import _lib
    import ctypes
    lib = _lib.load()
    for (w, h, o) in [(4000, 3000, 1), (4000, 3000, 6), (2049, 100, 1), (100, 2049, 8), (2048, 2048, 5), (5000, 3001, 7), (7, 9000, 2)]:
        ow, oh, rs = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        assert lib.ire_preprocess_plan(w, h, o, 2048, ctypes.byref(ow), ctypes.byref(oh), ctypes.byref(rs)) == 0
        assert (ow.value, oh.value, bool(rs.value)) == opp.plan(w, h, o)
    assert lib.ire_preprocess_plan(0, 10, 1, 2048, ctypes.byref(ow), ctypes.byref(oh), None) == 1      # IRE_ERR_INVALID_INPUT
    assert b"invalid" in lib.ire_last_error()
    assert lib.ire_preprocess_plan(10, 10, 9, 2048, ctypes.byref(ow), ctypes.byref(oh), None) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,o,max_dim", [(97, 131, 1, 64), (97, 131, 6, 64), (131, 97, 8, 50), (64, 200, 3, 100), (64, 200, 5, 100),
                                            (200, 64, 7, 33), (50, 70, 2, 2048), (50, 70, 4, 2048), (50, 70, 1, 2048), (301, 777, 6, 256)])
def test_engine_preprocess_bit_exact_vs_oracle(engine, h, w, o, max_dim):
    a = _img(h, w, seed=o)
    exp, _ = opp.preprocess_pixels(a, o, max_dim)
    got = engine.preprocess(a, orientation=o, max_dim=max_dim)
    assert got.shape == exp.shape and np.array_equal(got, exp)


@pytest.mark.gpu
def test_engine_preprocess_full_size_and_device_path(engine):
    import torch
    a = _img(3000, 4000, seed=5)                      # the reference's real case: 12 MP upload -> 2048 x 1536
    got = engine.preprocess(a, orientation=1)
    assert got.shape == (1536, 2048, 3)
    # size-independent checks at full size: rows/cols the oracle can afford + global statistics
    _, bh, th = opp.coefficients(4000, 2048)
    exp_rows = opp.resize(a[:64], 2048, 64)            # horizontal pass only on a strip
    mid = engine.preprocess_tensor(torch.from_numpy(np.ascontiguousarray(a[:64])).cuda(), max_dim=2048)   # 64x4000 -> fits: 33x2048
    assert mid.shape[1] == 2048
    dev = engine.preprocess_tensor(torch.from_numpy(a).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(dev.cpu().numpy(), got)
    assert abs(float(got.mean()) - float(a.mean())) < 0.5          # a normalised low-pass filter preserves the mean
    assert np.array_equal(engine.preprocess(a[:257, :300], orientation=6, max_dim=128), opp.preprocess_pixels(a[:257, :300], 6, 128)[0])
    assert exp_rows.shape == (64, 2048, 3)


@pytest.mark.gpu
def test_preprocess_image_round_trip(engine):
    from PIL import Image
    from image_restoration_platform_amd.preprocess import PreprocessError, preprocess_image
    a = _img(300, 500, seed=9)
    buf = io.BytesIO()
    im = Image.fromarray(a)
    ex = im.getexif()
    ex[0x0112] = 6
    im.save(buf, format="PNG", exif=ex)
    rec = preprocess_image(engine, buf.getvalue(), max_dim=256)
    assert rec["operations"] == ["auto_orient", "resize_256x154", "compress_jpeg_q85", "attach_sRGB_icc"]
    assert rec["original_metadata"] == {"width": 500, "height": 300, "format": "png", "orientation": 6}
    assert np.array_equal(rec["pixels"], opp.preprocess_pixels(a, 6, 256)[0])
    out = Image.open(io.BytesIO(rec["buffer"]))
    assert out.format == "JPEG" and out.size == (rec["processed_metadata"]["width"], rec["processed_metadata"]["height"])
    from PIL import ImageCms
    prof = ImageCms.ImageCmsProfile(io.BytesIO(out.info["icc_profile"]))              # 'attach_sRGB_icc' is what it says (imagePreprocess.js:65-68)
    assert "sRGB" in ImageCms.getProfileDescription(prof) and prof.profile.xcolor_space.strip() == "RGB"
    with pytest.raises(PreprocessError) as e:
        preprocess_image(engine, b"")
    assert e.value.status == 400
    with pytest.raises(PreprocessError) as e:
        preprocess_image(engine, b"definitely not an image")
    assert e.value.status == 422
