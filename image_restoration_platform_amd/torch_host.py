"""PyTorch-ROCm extension host of the engine (north_star: "server-python/FastAPI path calling the same engine via
PyTorch-ROCm extension"; SURVEY.md 8(b) caller 2).

`lib/_ire_torch.so` (csrc/torch_ext.cpp, built in-tree by build.py) is a real torch C++ extension: it takes and returns
torch tensors and calls the C ABI with their device pointers on the current torch HIP stream.  This module loads it,
fails loudly when it (or libire.so, or the GPU) is missing, and gives it the same method names as the ctypes host
(engine.Engine.*_tensor) so the FastAPI app can use either.  The ctypes host stays for callers without torch (numpy).
"""
import importlib.util
import os
import re

from . import _lib, weights as _weights
from .engine import EngineError

HERE = os.path.dirname(os.path.abspath(__file__))
EXT_PATH = os.path.join(HERE, "lib", "_ire_torch.so")
_ext = None


def load_extension():
    global _ext
    if _ext is None:
        if not os.path.exists(EXT_PATH):
            raise _lib.EngineLibraryMissing(f"service unavailable: {EXT_PATH} is missing -- run `python -m image_restoration_platform_amd.build`")
        import torch  # noqa: F401  (the extension links against libtorch)
        spec = importlib.util.spec_from_file_location("_ire_torch", EXT_PATH)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _ext = mod
    return _ext


def _raise(e):
    m = re.match(r"\[ire status (\d+)\] (.*)", str(e).split("\n")[0], flags=re.S)
    if m:
        raise EngineError(int(m.group(1)), m.group(2)) from None
    raise e


class TorchEngine:
    """Same engine, torch-extension front end: tensors in, tensors out, asynchronous on torch.cuda.current_stream()."""

    def __init__(self, device_index=0, max_batch=8, num_streams=0, weights_path="default", seed=0, precision="bf16"):
        self._ext = load_extension()
        if weights_path == "default":
            weights_path = _weights.ensure_default(seed)
        self._h = 0
        try:
            self._h = self._ext.init(_lib.LIB_PATH, device_index, max_batch, num_streams, weights_path or "", precision)
        except RuntimeError as e:
            _raise(e)

    def close(self):
        if getattr(self, "_h", 0):
            self._ext.shutdown(self._h)
            self._h = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _call(self, fn, *a):
        try:
            return fn(self._h, *a)
        except RuntimeError as e:
            _raise(e)

    def classify(self, rgb_u8, is_jpeg_u8=None):
        return self._call(self._ext.classify, rgb_u8, is_jpeg_u8)

    def restore(self, rgb_u8, scores=None, is_jpeg_u8=None):
        return self._call(self._ext.restore, rgb_u8, scores, is_jpeg_u8)

    def fuse(self, views_u8, noise_score=-1.0):
        return self._call(self._ext.fuse, views_u8, float(noise_score))

    def fuse_batch(self, views_u8, noise_scores=None):
        """views [S,k,H,W,3] uint8 cuda (S <= 16 view sets of one shape) -> (fused [S,H,W,3], shifts [S,k,2]); noise_scores: S floats (None / < 0: classify inside)."""
        import torch
        s = int(views_u8.shape[0])
        ns = torch.full((s,), -1.0, dtype=torch.float64) if noise_scores is None else torch.as_tensor([float(x) for x in noise_scores], dtype=torch.float64)
        return self._call(self._ext.fuse_batch, views_u8, ns)

    def preprocess(self, rgb_u8, orientation=1, max_dim=2048):
        """stored pixels [H,W,3] uint8 cuda -> upright pixels fitted inside max_dim (imagePreprocess.js:24-91, pixel part)."""
        return self._call(self._ext.preprocess, rgb_u8, int(orientation), int(max_dim))

    def encode_png_base64(self, rgb_u8):
        """[N,H,W,3] uint8 cuda -> [N, chars] uint8 cuda: the base64 text of each image's PNG file, written on the device."""
        return self._call(self._ext.encode_png_base64, rgb_u8)

    def restore_tiled(self, rgb_u8, nstrips, scores=None, is_jpeg_u8=None):
        return self._call(self._ext.restore_tiled, rgb_u8, int(nstrips), scores, is_jpeg_u8)
