"""The reference's five classifier fixtures rebuilt through REAL JPEG round trips (PIL's libjpeg codec), following
server-node/tests/utils/imageFixtures.js:5-45,91-93 step by step:

    createCleanImage / createBaseImage : 128x128 flat (180,180,180)          -> JPEG q95
    createBlurredImage                 : base -> decode -> blur(sigma 4)     -> JPEG q60
    createNoisyImage                   : uniform random bytes (seeded here;  -> JPEG q80
                                         the reference uses Math.random)
    createDarkImage                    : flat (10,10,10)                     -> JPEG q95
    createColorShiftedImage            : flat (220,80,40)                    -> JPEG q95

and decoded again to the RGB the classifier sees.  The codec is PIL's, not sharp's (sharp / libvips cannot be
installed here): quantisation tables follow libjpeg's defaults for those qualities, as sharp's do.  Chroma subsampling
does NOT match everywhere: PIL's default is 4:2:0 at every quality, sharp writes 4:4:4 from quality 90 up -- i.e. for the
three q95 fixtures.  Those three are flat fields (every 8x8 block is DC-only in all three planes), so the decoded pixels do not
depend on the subsampling; the q60 / q80 fixtures are 4:2:0 under both codecs.  The bytes are not claimed identical -- the
reference's assertions on these fixtures are inequalities (tests/classifierService.test.js:19-57), which is what they are
used for.
"""
import io

import numpy as np
from PIL import Image, ImageFilter

SIZE = 128


def _jpeg_roundtrip(rgb, quality):
    bio = io.BytesIO()
    Image.fromarray(rgb, "RGB").save(bio, format="JPEG", quality=quality)
    return np.ascontiguousarray(np.asarray(Image.open(io.BytesIO(bio.getvalue())).convert("RGB"), dtype=np.uint8))


def _flat(color):
    img = np.zeros((SIZE, SIZE, 3), np.uint8)
    img[:] = color
    return img


def base(color=(180, 180, 180)):
    return _jpeg_roundtrip(_flat(color), 95)


def blurred():
    b = Image.fromarray(base(), "RGB").filter(ImageFilter.GaussianBlur(4))
    return _jpeg_roundtrip(np.asarray(b, dtype=np.uint8), 60)


def noisy(seed=1234):
    data = np.random.default_rng(seed).integers(0, 256, (SIZE, SIZE, 3), dtype=np.uint8)
    return _jpeg_roundtrip(data, 80)


def dark():
    return base((10, 10, 10))


def color_shifted():
    return base((220, 80, 40))


def clean():
    return base()


# (name, builder, check(scores dict) -> bool): the assertions of classifierService.test.js:19-57
CASES = [
    ("blurred", blurred, lambda d: d["blur"] > 0.2 and d["noise"] >= 0 and "colorShift" in d),
    ("noisy", noisy, lambda d: d["noise"] > 0.3),
    ("dark", dark, lambda d: d["lowLight"] > 0.3),
    ("colorShifted", color_shifted, lambda d: d["colorShift"] > 0.25),
    ("clean", clean, lambda d: all(0 <= v <= 1 for v in d.values())),
]
