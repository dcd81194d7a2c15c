"""The C-ABI library loads and exports every symbol include/ire.h declares (no compute without a GPU),
and the product path fails loudly -- never falls back -- when there is no device."""
import ctypes
import os
import re

import pytest

from image_restoration_platform_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "ire.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ire_[a-z_0-9]+)\s*\(", txt)))


def test_header_and_binding_agree():
    assert _declared() == sorted(_lib.SYMBOLS.keys())


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name
    hdr = int(re.search(r"#define IRE_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "ire.h")).read()).group(1))
    assert _lib.load().ire_abi_version() == hdr == _lib.IRE_ABI_VERSION == 3


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the loud-failure path is exercised on CPU-only hosts")
    from image_restoration_platform_amd.engine import Engine, EngineError
    with pytest.raises(EngineError) as ei:
        Engine(weights_path=None)
    assert ei.value.status == _lib.IRE_ERR_UNAVAILABLE
    assert "service unavailable" in ei.value.message     # maps to SERVICE_UNAVAILABLE in _classifyError


def test_invalid_config_is_reported_not_crashed():
    lib = _lib.load()
    h = ctypes.c_void_p()
    cfg = _lib.IreConfig()
    cfg.struct_size = 4   # too small
    assert lib.ire_init(ctypes.byref(cfg), ctypes.byref(h)) == _lib.IRE_ERR_INVALID_INPUT
    assert b"invalid" in lib.ire_last_error()
    assert lib.ire_init(None, ctypes.byref(h)) == _lib.IRE_ERR_INVALID_INPUT
    cfg.struct_size = ctypes.sizeof(_lib.IreConfig)
    cfg.flags = 2         # an unknown flag bit (bit 0 = IRE_FLAG_RESULT_PNG_BASE64): rejected before any device is touched
    assert lib.ire_init(ctypes.byref(cfg), ctypes.byref(h)) == _lib.IRE_ERR_INVALID_INPUT
    assert b"flags" in lib.ire_last_error()
    assert lib.ire_profile_report(None, None, 0, None) == _lib.IRE_ERR_INVALID_INPUT
    # null engine handles are rejected, not dereferenced
    assert lib.ire_classify(None, None, 1, 8, 8, 24, None, None, None) == _lib.IRE_ERR_INVALID_INPUT
    assert lib.ire_profile_reset(None) == _lib.IRE_ERR_INVALID_INPUT


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "image_restoration_platform_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".js", ".mjs", ".cc")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "libire_oracle" not in src, f   # never loads the oracle library either


def _c_prototypes():
    """{name: (return kind, [argument kinds])} parsed from include/ire.h; kinds: int, double, size_t, void, string, ptr."""
    txt = open(os.path.join(ROOT, "include", "ire.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(int|void|size_t|const char\s*\*)\s+(ire_[a-z_0-9]+)\s*\(([^)]*)\)\s*;", txt):
        ret = "string" if "char" in m.group(1) else m.group(1)
        args = []
        for a in [x.strip() for x in m.group(3).split(",") if x.strip() and x.strip() != "void"]:
            if "*" in a:
                args.append("string" if re.match(r"const char\s*\*", a) else "ptr")
            elif a.startswith("double"):
                args.append("double")
            elif a.startswith("size_t"):
                args.append("size_t")
            else:
                args.append("int")
        out[m.group(2)] = (ret, args)
    return out


def test_ffi_napi_binding_declares_every_symbol_with_the_header_signature():
    """node/ire_ffi.mjs (the ffi-napi binding north_star names) cannot be loaded here (the module is not installable offline):
    parse its declaration table and hold it to include/ire.h, symbol by symbol."""
    src = open(os.path.join(ROOT, "image_restoration_platform_amd", "node", "ire_ffi.mjs")).read()
    decl = {}
    for m in re.finditer(r"^\s*(ire_[a-z_0-9]+):\s*\['(\w+)',\s*\[([^\]]*)\]\],?\s*$", src, flags=re.M):
        kinds = []
        for a in [x.strip() for x in m.group(3).split(",") if x.strip()]:
            kinds.append({"P": "ptr", "PP": "ptr", "IP": "ptr", "'int'": "int", "'double'": "double", "'size_t'": "size_t", "'string'": "string"}[a])
        decl[m.group(1)] = (m.group(2), kinds)
    protos = _c_prototypes()
    assert sorted(decl) == sorted(protos) == _declared()
    for name, (ret, args) in protos.items():
        assert decl[name] == (ret, args), (name, decl[name], (ret, args))
    assert int(re.search(r"IRE_ABI_VERSION = (\d+)", src).group(1)) == _lib.load().ire_abi_version()
