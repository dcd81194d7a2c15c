"""Python host binding of the engine's C ABI (include/ire.h) -- numpy and torch-tensor front ends.

This is the "PyTorch-ROCm extension" side of BASELINE.json's north_star: torch supplies device
memory and streams (tensor.data_ptr(), torch.cuda.current_stream()), the engine does the work.
Nothing here computes pixels on the CPU; if libire.so or the GPU is missing every call raises.
"""
import ctypes

import numpy as np

from . import _lib, weights as _weights

KEYS = ["blur", "noise", "lowLight", "compression", "scratch", "fade", "colorShift"]  # classifier.js:62-70
FAMILIES = ["classifier", "conv3x3", "conv1x1", "stem", "head", "gn_finalize", "fusion", "all"]


class EngineError(RuntimeError):
    """Raised for a non-zero ire_status; .message feeds RestoratorService._classifyError
    (restorator.js:241-265): it contains 'invalid' / 'timeout' / 'service unavailable'."""

    def __init__(self, status, message):
        super().__init__(message)
        self.status = status
        self.message = message
        self.code = {1: "ENGINE_INVALID_INPUT", 2: "ENGINE_TIMEOUT", 3: "ENGINE_UNAVAILABLE"}.get(status, "ENGINE_INTERNAL")


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else None


class Engine:
    def __init__(self, device_index=0, max_batch=8, num_streams=0, weights_path="default", seed=0, flags=0, precision="bf16"):
        self._lib = _lib.load()
        if weights_path == "default":
            weights_path = _weights.ensure_default(seed)
        cfg = _lib.IreConfig()
        cfg.struct_size = ctypes.sizeof(_lib.IreConfig)
        cfg.device_index = device_index
        if precision not in ("bf16", "fp8"):
            raise EngineError(1, "invalid precision: 'bf16' or 'fp8'")
        cfg.precision = 1 if precision == "fp8" else 0          # IRE_PRECISION_*
        cfg.max_batch = max_batch
        cfg.num_streams = num_streams
        cfg.weights_path = weights_path.encode() if weights_path else None
        cfg.flags = flags
        self._flags = flags
        h = ctypes.c_void_p()
        self._h = None
        self._check(self._lib.ire_init(ctypes.byref(cfg), ctypes.byref(h)))
        self._h = h
        self.max_batch = max_batch

    def load_weights(self, blob):
        """RestoreNet-v0 weights from memory (the bytes of a weight file): ire_load_weights."""
        b = bytes(blob)
        self._check(self._lib.ire_load_weights(self._h, b, len(b)))

    def _check(self, rc):
        if rc != 0:
            msg = self._lib.ire_last_error()
            raise EngineError(rc, msg.decode() if msg else f"engine error {rc}")

    def close(self):
        if getattr(self, "_h", None):
            for ref in getattr(self, "_sessions", []):     # open strip sessions die with their engine (ire_shutdown would only
                s = ref()                                  # leave them as empty shells): close them first
                if s is not None:
                    s.close()
            self._sessions = []
            self._lib.ire_shutdown(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host (numpy) -------------------------------------------------------------------------
    @staticmethod
    def _as_batch(rgb):
        rgb = np.asarray(rgb)
        if rgb.dtype != np.uint8:
            raise EngineError(1, "invalid input: image dtype must be uint8")
        if rgb.ndim == 3:
            rgb = rgb[None]
        if rgb.ndim != 4 or rgb.shape[3] != 3:
            raise EngineError(1, "invalid input: expected [N,H,W,3] uint8")
        return np.ascontiguousarray(rgb)

    def classify(self, rgb, is_jpeg=True):
        """-> (scores [N,7] float64, labels [N] int32).  ClassifierService.analyze (classifier.js:40)."""
        rgb = self._as_batch(rgb)
        n, h, w, _ = rgb.shape
        jp = np.ascontiguousarray(np.broadcast_to(np.asarray(is_jpeg, dtype=np.uint8), (n,)))
        scores = np.zeros((n, 7), np.float64)
        labels = np.zeros(n, np.int32)
        self._check(self._lib.ire_classify(self._h, _ptr(rgb), n, h, w, 3 * w, _ptr(jp), _ptr(scores), _ptr(labels)))
        return scores, labels

    def restore(self, rgb, scores=None, is_jpeg=True, return_timings=False):
        """-> restored [N,H,W,3] uint8.  The step behind GeminiClient.restoreImage (geminiClient.js:32)."""
        rgb = self._as_batch(rgb)
        n, h, w, _ = rgb.shape
        jp = np.ascontiguousarray(np.broadcast_to(np.asarray(is_jpeg, dtype=np.uint8), (n,)))
        sc = None
        if scores is not None:
            sc = np.ascontiguousarray(np.asarray(scores, dtype=np.float64).reshape(n, 7))
        out = np.empty_like(rgb)
        t = _lib.IreTimings()
        self._check(self._lib.ire_restore(self._h, _ptr(rgb), n, h, w, _ptr(sc), _ptr(jp), _ptr(out), ctypes.byref(t)))
        if return_timings:
            return out, {"classify_ms": t.classify_ms, "restore_ms": t.restore_ms, "total_ms": t.total_ms}
        return out

    def fuse(self, views, noise_score=-1.0):
        """views [k,H,W,3] uint8, k in 2..3 -> (fused [H,W,3], shifts [k,2] (dy,dx))."""
        views = self._as_batch(views)
        k, h, w, _ = views.shape
        out = np.empty((h, w, 3), np.uint8)
        shifts = np.zeros((k, 2), np.int32)
        t = _lib.IreTimings()
        self._check(self._lib.ire_fuse(self._h, _ptr(views), k, h, w, float(noise_score), _ptr(out), _ptr(shifts),
                                       ctypes.byref(t)))
        return out, shifts

    def preprocess_plan(self, width, height, orientation=1, max_dim=2048):
        """-> (out_w, out_h, resized): size rule of imagePreprocess.js:12-22,46-55 (host arithmetic, no GPU work)."""
        ow, oh, rs = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        self._check(self._lib.ire_preprocess_plan(int(width), int(height), int(orientation), int(max_dim), ctypes.byref(ow),
                                                  ctypes.byref(oh), ctypes.byref(rs)))
        return ow.value, oh.value, bool(rs.value)

    def preprocess(self, rgb, orientation=1, max_dim=2048):
        """stored pixels [H,W,3] uint8 -> upright pixels fitted inside max_dim (EXIF orient + Lanczos-3), on the GPU."""
        rgb = np.ascontiguousarray(rgb)
        if rgb.dtype != np.uint8 or rgb.ndim != 3 or rgb.shape[2] != 3:
            raise EngineError(1, "invalid input: expected [H,W,3] uint8")
        h, w, _ = rgb.shape
        ow, oh, _ = self.preprocess_plan(w, h, orientation, max_dim)
        out = np.empty((oh, ow, 3), np.uint8)
        self._check(self._lib.ire_preprocess(self._h, _ptr(rgb), h, w, int(orientation), int(max_dim), _ptr(out), oh, ow))
        return out

    def preprocess_tensor(self, rgb_u8, orientation=1, max_dim=2048, stream=None):
        import torch
        assert rgb_u8.is_cuda and rgb_u8.dtype == torch.uint8 and rgb_u8.is_contiguous() and rgb_u8.dim() == 3
        h, w, _ = rgb_u8.shape
        ow, oh, _ = self.preprocess_plan(w, h, orientation, max_dim)
        out = torch.empty((oh, ow, 3), dtype=torch.uint8, device=rgb_u8.device)
        self._check(self._lib.ire_preprocess_device(self._h, ctypes.c_void_p(rgb_u8.data_ptr()), h, w, int(orientation), int(max_dim),
                                                    ctypes.c_void_p(out.data_ptr()), oh, ow, self._stream_ptr(stream)))
        return out

    def submit(self, rgb, is_jpeg=True, scores=None):
        """Queue one image with the engine's batcher (in-flight jobs of one shape coalesce into engine batches);
        scores: the 7 scores a previous classify() returned for this image => it is not classified again."""
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        h, w, _ = rgb.shape
        sc = None if scores is None else np.ascontiguousarray(np.asarray(scores, dtype=np.float64).reshape(7))
        job = ctypes.c_void_p()
        self._check(self._lib.ire_submit(self._h, _ptr(rgb), h, w, int(bool(is_jpeg)), _ptr(sc), ctypes.byref(job)))
        return (job, h, w)

    def stats(self):
        """Service gauges (f4: getHealthStatus / the /health/ready dependency entry)."""
        st = _lib.IreEngineStats()
        st.struct_size = ctypes.sizeof(_lib.IreEngineStats)
        self._check(self._lib.ire_get_stats(self._h, ctypes.byref(st)))
        return {"queueDepth": st.queue_depth, "batches": st.batches, "images": st.images, "lastBatch": st.last_batch,
                "maxBatch": st.max_batch, "imagesPerSec": st.images_per_sec}

    def max_batch_for(self, h, w):
        return int(self._lib.ire_max_batch_for(self._h, int(h), int(w)))

    def poll(self, job, timeout_ms=-1):
        """-> (restored [H,W,3] uint8 -- or, for an engine created with flags=IRE_FLAG_RESULT_PNG_BASE64, the `bytes` of the base64 text
        of its PNG file --, scores[7], timings)"""
        handle, h, w = job
        text = bool(getattr(self, "_flags", 0) & _lib.IRE_FLAG_RESULT_PNG_BASE64)
        out = np.empty(self.png_base64_bytes(h, w), np.uint8) if text else np.empty((h, w, 3), np.uint8)
        scores = np.zeros(7, np.float64)
        t = _lib.IreTimings()
        self._check(self._lib.ire_poll(self._h, handle, timeout_ms, _ptr(out), _ptr(scores), ctypes.byref(t)))
        if text:
            out = out.tobytes()
        return out, scores, {"classify_ms": t.classify_ms, "restore_ms": t.restore_ms, "total_ms": t.total_ms}

    def png_base64_bytes(self, h, w):
        return int(self._lib.ire_png_base64_bytes(int(h), int(w)))

    def encode_png_base64(self, rgb):
        """[N,H,W,3] (or [H,W,3]) uint8 -> list of N `bytes` (one for a single image): the base64 text of a PNG file of each image,
        encoded on the device (stored deflate blocks: every PNG decoder reads it)."""
        single = np.asarray(rgb).ndim == 3
        x = self._as_batch(rgb)
        n, h, w, _ = x.shape
        cb = self.png_base64_bytes(h, w)
        if cb == 0:
            raise EngineError(_lib.IRE_ERR_INVALID_INPUT, "invalid image size for the PNG encoder: width must be a multiple of 8")
        stride = (cb + 15) // 16 * 16
        out = np.empty((n, stride), np.uint8)
        self._check(self._lib.ire_encode_png_base64(self._h, _ptr(x), n, h, w, _ptr(out), stride))
        res = [out[i, :cb].tobytes() for i in range(n)]
        return res[0] if single else res

    def encode_png_base64_tensor(self, rgb_u8, stream=None):
        """cuda uint8 [N,H,W,3] -> cuda uint8 [N, chars] ASCII (asynchronous on the stream)."""
        import torch
        n, h, w, _ = rgb_u8.shape
        cb = self.png_base64_bytes(h, w)
        out = torch.empty((n, cb), dtype=torch.uint8, device=rgb_u8.device)
        self._check(self._lib.ire_encode_png_base64_device(self._h, ctypes.c_void_p(rgb_u8.data_ptr()), n, h, w, ctypes.c_void_p(out.data_ptr()), cb,
                                                           self._stream_ptr(stream)))
        return out

    def release(self, job):
        """Give a submitted job up without fetching it (after a poll() timeout the caller will not repeat): ire_job_release."""
        handle, _, _ = job
        self._check(self._lib.ire_job_release(self._h, handle))

    def affinity(self):
        """(cpulist, numa_node) the engine's service threads are bound to ("" / -1: no binding)."""
        buf = ctypes.create_string_buffer(4096)
        node = ctypes.c_int32(-1)
        self._check(self._lib.ire_engine_affinity(self._h, buf, len(buf), ctypes.byref(node)))
        return buf.value.decode(), int(node.value)

    # ---- device (torch tensors) ---------------------------------------------------------------
    @staticmethod
    def _stream_ptr(stream):
        import torch
        s = stream if stream is not None else torch.cuda.current_stream()
        return ctypes.c_void_p(s.cuda_stream)

    def classify_tensor(self, rgb_u8, is_jpeg_u8=None, stream=None):
        """rgb_u8: cuda uint8 [N,H,W,3] contiguous -> (scores cuda float64 [N,7], labels cuda int32 [N])."""
        import torch
        assert rgb_u8.is_cuda and rgb_u8.dtype == torch.uint8 and rgb_u8.is_contiguous() and rgb_u8.dim() == 4
        n, h, w, _ = rgb_u8.shape
        scores = torch.empty((n, 7), dtype=torch.float64, device=rgb_u8.device)
        labels = torch.empty((n,), dtype=torch.int32, device=rgb_u8.device)
        jp = ctypes.c_void_p(is_jpeg_u8.data_ptr()) if is_jpeg_u8 is not None else None
        self._check(self._lib.ire_classify_device(self._h, ctypes.c_void_p(rgb_u8.data_ptr()), n, h, w, jp,
                                                  ctypes.c_void_p(scores.data_ptr()), ctypes.c_void_p(labels.data_ptr()),
                                                  self._stream_ptr(stream)))
        return scores, labels

    def restore_tensor(self, rgb_u8, out_u8=None, scores=None, is_jpeg_u8=None, stream=None):
        """Asynchronous on the torch stream: classify (unless scores given) + RestoreNet-v0."""
        import torch
        assert rgb_u8.is_cuda and rgb_u8.dtype == torch.uint8 and rgb_u8.is_contiguous() and rgb_u8.dim() == 4
        n, h, w, _ = rgb_u8.shape
        if out_u8 is None:
            out_u8 = torch.empty_like(rgb_u8)
        sc = ctypes.c_void_p(scores.data_ptr()) if scores is not None else None
        jp = ctypes.c_void_p(is_jpeg_u8.data_ptr()) if is_jpeg_u8 is not None else None
        self._check(self._lib.ire_restore_device(self._h, ctypes.c_void_p(rgb_u8.data_ptr()), n, h, w, sc, jp,
                                                 ctypes.c_void_p(out_u8.data_ptr()), self._stream_ptr(stream)))
        return out_u8

    def fuse_tensor(self, views_u8, noise_score=-1.0, stream=None):
        import torch
        assert views_u8.is_cuda and views_u8.dtype == torch.uint8 and views_u8.is_contiguous() and views_u8.dim() == 4
        k, h, w, _ = views_u8.shape
        out = torch.empty((h, w, 3), dtype=torch.uint8, device=views_u8.device)
        shifts = torch.zeros((k, 2), dtype=torch.int32, device=views_u8.device)
        self._check(self._lib.ire_fuse_device(self._h, ctypes.c_void_p(views_u8.data_ptr()), k, h, w, float(noise_score),
                                              ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(shifts.data_ptr()),
                                              self._stream_ptr(stream)))
        return out, shifts

    def fuse_batch_tensor(self, views_u8, noise_scores=None, stream=None):
        """views [B,k,H,W,3] uint8 on the GPU (B <= 16 view sets of one shape) -> (fused [B,H,W,3], shifts [B,k,2]).
        noise_scores: B floats (host), each < 0 / None => classify view 0 of that set inside.  One pass of the kernel chain."""
        import torch
        assert views_u8.is_cuda and views_u8.dtype == torch.uint8 and views_u8.is_contiguous() and views_u8.dim() == 5
        b, k, h, w, _ = views_u8.shape
        ns = (ctypes.c_double * b)(*([-1.0] * b if noise_scores is None else [float(x) for x in noise_scores]))
        out = torch.empty((b, h, w, 3), dtype=torch.uint8, device=views_u8.device)
        shifts = torch.zeros((b, k, 2), dtype=torch.int32, device=views_u8.device)
        self._check(self._lib.ire_fuse_batch_device(self._h, ctypes.c_void_p(views_u8.data_ptr()), b, k, h, w, ns,
                                                    ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(shifts.data_ptr()),
                                                    self._stream_ptr(stream)))
        return out, shifts

    # ---- cfg 4: row strips ----------------------------------------------------------------------
    def restore_tiled_tensor(self, rgb_u8, nstrips, out_u8=None, scores=None, is_jpeg_u8=None, stream=None):
        """One [H,W,3] image restored as `nstrips` row strips on this GPU (virtual ranks: per-level halo exchange and the
        GroupNorm partials gather are in-device copies).  Bit-identical to restore_tensor on the whole image."""
        import torch
        assert rgb_u8.is_cuda and rgb_u8.dtype == torch.uint8 and rgb_u8.is_contiguous() and rgb_u8.dim() == 3
        h, w, _ = rgb_u8.shape
        if out_u8 is None:
            out_u8 = torch.empty_like(rgb_u8)
        sc = ctypes.c_void_p(scores.data_ptr()) if scores is not None else None
        jp = ctypes.c_void_p(is_jpeg_u8.data_ptr()) if is_jpeg_u8 is not None else None
        self._check(self._lib.ire_restore_tiled_device(self._h, ctypes.c_void_p(rgb_u8.data_ptr()), h, w, int(nstrips), sc, jp,
                                                       ctypes.c_void_p(out_u8.data_ptr()), self._stream_ptr(stream)))
        return out_u8

    def open_strips(self, h, w, nstrips_total, first_strip, nlocal=1):
        import weakref
        s = StripSession(self, h, w, nstrips_total, first_strip, nlocal)
        if not hasattr(self, "_sessions"):
            self._sessions = []
        self._sessions = [r for r in self._sessions if r() is not None] + [weakref.ref(s)]
        return s

    # ---- diagnostics --------------------------------------------------------------------------
    def classifier_sums(self, n):
        sums = np.zeros((n, 14), np.uint64)
        self._check(self._lib.ire_debug_classifier_sums(self._h, n, _ptr(sums)))
        return sums

    def debug_capture(self, on=True):
        self._check(self._lib.ire_debug_capture(self._h, int(on)))

    def activation(self, name):
        cnt = ctypes.c_size_t(0)
        self._check(self._lib.ire_debug_activation(self._h, name.encode(), None, ctypes.byref(cnt)))
        out = np.zeros(cnt.value, np.float32)
        self._check(self._lib.ire_debug_activation(self._h, name.encode(), _ptr(out), ctypes.byref(cnt)))
        return out

    def profile_enable(self, mode=1):
        """Low byte: 0 off, 1 every kernel family, 2 the 3x3 conv family only (fewest HIP events); `mode | (N << 8)`: in mode 2
        bracket the launches of every N-th pass of the network only (an event record is a packet between two kernels)."""
        self._check(self._lib.ire_profile_enable(self._h, int(mode)))

    def profile_reset(self):
        self._check(self._lib.ire_profile_reset(self._h))

    def profile_query(self, family="all"):
        ms, n, fl, by = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
        self._check(self._lib.ire_profile_query(self._h, family.encode(), ctypes.byref(ms), ctypes.byref(n),
                                                ctypes.byref(fl), ctypes.byref(by)))
        return {"ms": ms.value, "launches": n.value, "flops": fl.value, "bytes": by.value}

    def profile_report(self):
        """Per-layer-group rows of the profiled convolution launches since the last reset (ire_profile_report)."""
        import json
        need = ctypes.c_size_t(0)
        self._check(self._lib.ire_profile_report(self._h, None, 0, ctypes.byref(need)))
        buf = ctypes.create_string_buffer(need.value + 64)
        self._check(self._lib.ire_profile_report(self._h, buf, len(buf), ctypes.byref(need)))
        return json.loads(buf.value.decode())


class StripSession:
    """One rank's strips of a tiled restoration (include/ire.h "cfg 4"): the layer program runs one op at a time and the host
    moves halo rows / GroupNorm partials between ranks in between (tiled.py).  All buffers the ranks exchange are torch tensors
    owned here: `stats` (the global partials array, all-gathered in place) and four halo staging rows."""

    def __init__(self, engine, h, w, nstrips_total, first_strip, nlocal=1):
        import torch
        self.eng, self._lib = engine, engine._lib
        self.h, self.w, self.total, self.first, self.nlocal = h, w, nstrips_total, first_strip, nlocal
        self.rows = (h // nstrips_total) * nlocal
        dev = torch.device("cuda", torch.cuda.current_device())
        self.stats = torch.zeros(int(self._lib.ire_strips_stats_bytes(h, w)), dtype=torch.uint8, device=dev)
        row_max = w * 32 * 2                       # W_l * C_l * 2 bytes is the same at every level
        self.send_up, self.send_down, self.recv_up, self.recv_down = (torch.zeros(row_max, dtype=torch.uint8, device=dev) for _ in range(4))
        hnd = ctypes.c_void_p()
        self._s = None
        engine._check(self._lib.ire_strips_open(engine._h, h, w, nstrips_total, first_strip, nlocal, ctypes.c_void_p(self.stats.data_ptr()),
                                                ctypes.byref(hnd)))
        self._s = hnd
        self.num_ops = int(self._lib.ire_strips_num_ops(self._s))

    def close(self):
        if getattr(self, "_s", None):
            self._lib.ire_strips_close(self._s)
            self._s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_input(self, rows_with_halo_u8, scores_f64, stream=None):
        assert rows_with_halo_u8.is_cuda and rows_with_halo_u8.is_contiguous() and tuple(rows_with_halo_u8.shape) == (self.rows + 2, self.w, 3)
        self.eng._check(self._lib.ire_strips_set_input(self._s, ctypes.c_void_p(rows_with_halo_u8.data_ptr()), ctypes.c_void_p(scores_f64.data_ptr()),
                                                       Engine._stream_ptr(stream)))

    def run_op(self, k, stream=None):
        info = _lib.IreStripXchg()
        self.eng._check(self._lib.ire_strips_run_op(self._s, int(k), Engine._stream_ptr(stream), ctypes.byref(info)))
        return info

    def pack_halo(self, k, stream=None):
        self.eng._check(self._lib.ire_strips_pack_halo(self._s, int(k), ctypes.c_void_p(self.send_up.data_ptr()), ctypes.c_void_p(self.send_down.data_ptr()),
                                                       Engine._stream_ptr(stream)))

    def unpack_halo(self, k, stream=None):
        self.eng._check(self._lib.ire_strips_unpack_halo(self._s, int(k), ctypes.c_void_p(self.recv_up.data_ptr()), ctypes.c_void_p(self.recv_down.data_ptr()),
                                                         Engine._stream_ptr(stream)))

    def get_output(self, stream=None):
        import torch
        out = torch.empty((self.rows, self.w, 3), dtype=torch.uint8, device=self.stats.device)
        self.eng._check(self._lib.ire_strips_get_output(self._s, ctypes.c_void_p(out.data_ptr()), Engine._stream_ptr(stream)))
        return out
