"""ctypes view of oracle/classifier_oracle.c (TEST INFRASTRUCTURE ONLY).

Follows server-node/src/services/classifier.js:40-337; see the C file header for
the pinning status of each libvips assumption.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libire_oracle.so")
KEYS = ["blur", "noise", "lowLight", "compression", "scratch", "fade", "colorShift"]


class Sums(ctypes.Structure):
    _fields_ = [
        ("sum_c", ctypes.c_uint64 * 3), ("sumsq_c", ctypes.c_uint64 * 3),
        ("sum_blur", ctypes.c_uint64), ("sumsq_blur", ctypes.c_uint64),
        ("sum_e8", ctypes.c_uint64), ("sumsq_e8", ctypes.c_uint64),
        ("sum_e9", ctypes.c_uint64), ("sumsq_e9", ctypes.c_uint64),
        ("scratch_v", ctypes.c_uint64), ("scratch_h", ctypes.c_uint64),
    ]

    def as_list(self):
        return (list(self.sum_c) + list(self.sumsq_c) + [self.sum_blur, self.sumsq_blur, self.sum_e8,
                self.sumsq_e8, self.sum_e9, self.sumsq_e9, self.scratch_v, self.scratch_h])


def build(force=False):
    src = os.path.join(_HERE, "classifier_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libire_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        u8p = ctypes.POINTER(ctypes.c_uint8)
        _lib.ire_oracle_classify.argtypes = [u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32),
                                             ctypes.POINTER(Sums)]
        _lib.ire_oracle_classify_twopass.argtypes = _lib.ire_oracle_classify.argtypes[:7]
        _lib.ire_oracle_planes.argtypes = [u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int] + [u8p] * 5
        _lib.ire_oracle_grey_tables.argtypes = [ctypes.POINTER(ctypes.c_uint32)] * 2
    return _lib


def _u8(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def classify(rgb, is_jpeg=True, twopass=False, with_sums=False):
    """rgb: HxWx3 uint8 (C-contiguous).  Returns (scores[7] float64, label[, sums])."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    if rgb.ndim != 3 or rgb.shape[2] != 3:
        raise ValueError("rgb must be HxWx3")
    h, w, _ = rgb.shape
    scores = np.zeros(7, dtype=np.float64)
    label = ctypes.c_int32(-1)
    sp = scores.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    if twopass:
        rc = lib().ire_oracle_classify_twopass(_u8(rgb), h, w, 3 * w, int(bool(is_jpeg)), sp, ctypes.byref(label))
        sums = None
    else:
        sums = Sums()
        rc = lib().ire_oracle_classify(_u8(rgb), h, w, 3 * w, int(bool(is_jpeg)), sp, ctypes.byref(label),
                                       ctypes.byref(sums))
    if rc != 0:
        raise ValueError("oracle: invalid input")
    if with_sums:
        return scores, int(label.value), sums
    return scores, int(label.value)


def planes(rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, _ = rgb.shape
    out = {k: np.zeros((h, w), np.uint8) for k in ("grey", "e8", "e9", "e4")}
    out["blur"] = np.zeros((h, w, 3), np.uint8)
    rc = lib().ire_oracle_planes(_u8(rgb), h, w, 3 * w, _u8(out["grey"]), _u8(out["e8"]), _u8(out["e9"]),
                                 _u8(out["e4"]), _u8(out["blur"]))
    if rc != 0:
        raise ValueError("oracle: invalid input")
    return out


def grey_tables():
    lin = np.zeros(256, np.uint32)
    thr = np.zeros(256, np.uint32)
    lib().ire_oracle_grey_tables(lin.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)),
                                 thr.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)))
    return lin, thr
