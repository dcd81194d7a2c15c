"""ctypes loader for libire.so -- fails loudly when the HIP library is missing (no CPU fallback)."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IRE_LIB") or os.path.join(HERE, "lib", "libire.so")   # IRE_LIB: A/B another build of the same ABI (tools/ab_bench.sh)

IRE_OK, IRE_ERR_INVALID_INPUT, IRE_ERR_TIMEOUT, IRE_ERR_UNAVAILABLE, IRE_ERR_INTERNAL = range(5)
IRE_FLAG_RESULT_PNG_BASE64 = 1
IRE_ABI_VERSION = 3      # include/ire.h; load() refuses a library of another version (tests/test_abi.py cross-checks the three copies)


class IreConfig(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("device_index", ctypes.c_int32), ("precision", ctypes.c_int32),
                ("max_batch", ctypes.c_int32), ("num_streams", ctypes.c_int32), ("weights_path", ctypes.c_char_p),
                ("flags", ctypes.c_uint32)]


class IreTimings(ctypes.Structure):
    _fields_ = [("classify_ms", ctypes.c_double), ("restore_ms", ctypes.c_double), ("total_ms", ctypes.c_double)]


class IreEngineStats(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("queue_depth", ctypes.c_int32), ("batches", ctypes.c_int64),
                ("images", ctypes.c_int64), ("last_batch", ctypes.c_int32), ("max_batch", ctypes.c_int32),
                ("images_per_sec", ctypes.c_double)]


class IreStripXchg(ctypes.Structure):
    _fields_ = [("halo_bytes", ctypes.c_int32), ("has_up", ctypes.c_int32), ("has_down", ctypes.c_int32),
                ("stats_offset_bytes", ctypes.c_int64), ("stats_local_bytes", ctypes.c_int64), ("stats_total_bytes", ctypes.c_int64)]


# every symbol include/ire.h declares: (name, restype, argtypes)
_vp, _i, _u8p = ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p
SYMBOLS = {
    "ire_abi_version": (_i, []),
    "ire_init": (_i, [ctypes.POINTER(IreConfig), ctypes.POINTER(_vp)]),
    "ire_shutdown": (None, [_vp]),
    "ire_last_error": (ctypes.c_char_p, []),
    "ire_load_weights": (_i, [_vp, _vp, ctypes.c_size_t]),
    "ire_max_batch_for": (_i, [_vp, _i, _i]),
    "ire_classify": (_i, [_vp, _u8p, _i, _i, _i, _i, _u8p, _vp, _vp]),
    "ire_restore": (_i, [_vp, _u8p, _i, _i, _i, _vp, _u8p, _u8p, ctypes.POINTER(IreTimings)]),
    "ire_fuse": (_i, [_vp, _u8p, _i, _i, _i, ctypes.c_double, _u8p, _vp, ctypes.POINTER(IreTimings)]),
    "ire_classify_device": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "ire_restore_device": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "ire_fuse_device": (_i, [_vp, _vp, _i, _i, _i, ctypes.c_double, _vp, _vp, _vp]),
    "ire_fuse_batch_device": (_i, [_vp, _vp, _i, _i, _i, _i, ctypes.POINTER(ctypes.c_double), _vp, _vp, _vp]),
    "ire_preprocess_plan": (_i, [_i, _i, _i, _i, ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    "ire_preprocess": (_i, [_vp, _u8p, _i, _i, _i, _i, _u8p, _i, _i]),
    "ire_preprocess_device": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    "ire_png_base64_bytes": (ctypes.c_size_t, [_i, _i]),
    "ire_encode_png_base64_device": (_i, [_vp, _vp, _i, _i, _i, _vp, ctypes.c_size_t, _vp]),
    "ire_encode_png_base64": (_i, [_vp, _u8p, _i, _i, _i, _u8p, ctypes.c_size_t]),
    "ire_submit": (_i, [_vp, _u8p, _i, _i, _i, _vp, ctypes.POINTER(_vp)]),
    "ire_restore_tiled_device": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "ire_strips_stats_bytes": (ctypes.c_size_t, [_i, _i]),
    "ire_strips_open": (_i, [_vp, _i, _i, _i, _i, _i, _vp, ctypes.POINTER(_vp)]),
    "ire_strips_close": (None, [_vp]),
    "ire_strips_num_ops": (_i, [_vp]),
    "ire_strips_set_input": (_i, [_vp, _vp, _vp, _vp]),
    "ire_strips_run_op": (_i, [_vp, _i, _vp, ctypes.POINTER(IreStripXchg)]),
    "ire_strips_pack_halo": (_i, [_vp, _i, _vp, _vp, _vp]),
    "ire_strips_unpack_halo": (_i, [_vp, _i, _vp, _vp, _vp]),
    "ire_strips_get_output": (_i, [_vp, _vp, _vp]),
    "ire_get_stats": (_i, [_vp, ctypes.POINTER(IreEngineStats)]),
    "ire_poll": (_i, [_vp, _vp, _i, _u8p, _vp, ctypes.POINTER(IreTimings)]),
    "ire_job_release": (_i, [_vp, _vp]),
    "ire_affinity_plan": (_i, [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int32),
                               ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "ire_engine_affinity": (_i, [_vp, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int32)]),
    "ire_debug_classifier_sums": (_i, [_vp, _i, _vp]),
    "ire_debug_capture": (_i, [_vp, _i]),
    "ire_debug_activation": (_i, [_vp, ctypes.c_char_p, _vp, ctypes.POINTER(ctypes.c_size_t)]),
    "ire_profile_enable": (_i, [_vp, _i]),
    "ire_profile_query": (_i, [_vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64),
                               ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "ire_profile_reset": (_i, [_vp]),
    "ire_profile_report": (_i, [_vp, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]),
}

_lib = None


class EngineLibraryMissing(RuntimeError):
    pass


def load():
    """Load libire.so; raise (never fall back) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineLibraryMissing(
            f"service unavailable: {LIB_PATH} is missing -- run `python -m image_restoration_platform_amd.build` "
            "(the engine has no CPU fallback)")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError => the library does not export what ire.h declares
        fn.restype = res
        fn.argtypes = args
    if lib.ire_abi_version() != IRE_ABI_VERSION:
        raise EngineLibraryMissing(f"service unavailable: {LIB_PATH} has ABI version {lib.ire_abi_version()}, this binding needs {IRE_ABI_VERSION} "
                                   "-- rebuild with `python -m image_restoration_platform_amd.build`")
    _lib = lib
    return lib
