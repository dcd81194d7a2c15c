"""CPU oracle for RestoreNet-v0 (PyTorch-CPU fp32) -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED against the reference: the reference's restoration step is a remote call
(server-node/src/clients/geminiClient.js:32-97, called from restorator.js:88-94) to a proprietary
model, no test pins its pixels (tests/restoratorService.test.js:20-28 mocks it), so there is no
reference output to compare with.  This file is the independent fp32 statement of the
BUILD-DEFINED network (architecture: image_restoration_platform_amd/weights.py docstring,
SURVEY.md Appendix C); the HIP engine (bf16 storage, fp32 accumulate) is checked against it
within the tolerance stated in tests/test_restore_gpu.py.

Two modes:
  restore(...)                fp32 everywhere (the "true" function).
  restore(..., emulate_bf16=True)
                              rounds activations to bf16 at the engine's storage points (conv
                              outputs, residual sums, the activated conv inputs) so that a layer-
                              by-layer comparison only sees accumulation-order noise -- used to
                              localise indexing bugs that an end-to-end tolerance could hide.
  restore(..., emulate_bf16=True, emulate_fp8=True)
                              additionally quantises the two MFMA operands of every C >= 128 ResBlock
                              convolution the way IRE_PRECISION_FP8 does (csrc/conv_w4.hip, engine.cpp):
                              weights to OCP e4m3 with one scale per output channel (max|w| -> 448),
                              the activated input to e4m3 after a x16 scale, clamped at 448.
"""
import numpy as np
import torch
import torch.nn.functional as F

WIDTHS = (32, 64, 128, 256)
FILM_OFFSETS = (0, 64, 192, 448)
GN_EPS = 1e-5


def _bf16(t):
    return t.to(torch.bfloat16).to(torch.float32)


FP8_ACT_SCALE = 16.0


def _e4m3(t):
    return torch.clamp(t, -448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32)


class _Net:
    def __init__(self, weights, emulate_bf16=False, capture=None, emulate_fp8=False):
        self.w = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in weights.items()}
        self.emu = emulate_bf16
        self.fp8 = emulate_fp8
        self.cap = capture

    def q(self, t):
        return _bf16(t) if self.emu else t

    def keep(self, name, t):  # t: NCHW -> store NHWC float32
        if self.cap is not None:
            self.cap[name] = t.permute(0, 2, 3, 1).contiguous().numpy().copy()

    def gn_film_silu(self, x, prefix, level, film):
        c = x.shape[1]
        g, b = self.w[prefix + ".g"], self.w[prefix + ".b"]
        n = x.shape[0]
        xg = x.reshape(n, 8, -1).double()
        mean = xg.mean(dim=2)
        var = (xg * xg).mean(dim=2) - mean * mean
        var = torch.clamp(var, min=0.0)
        rstd = 1.0 / torch.sqrt(var + GN_EPS)
        mean = mean.float().repeat_interleave(c // 8, dim=1)  # [n, c]
        rstd = rstd.float().repeat_interleave(c // 8, dim=1)
        off = FILM_OFFSETS[level]
        s = film[:, off:off + c]
        t = film[:, off + c:off + 2 * c]
        rg = rstd * g[None, :]
        a = rg * (1.0 + s)
        bb = (b[None, :] - mean * rg) * (1.0 + s) + t
        y = x * a[:, :, None, None] + bb[:, :, None, None]
        if self.fp8 and c >= 128:          # the activated operand goes to e4m3 (x16), not to bf16
            return _e4m3(F.silu(y) * FP8_ACT_SCALE) / FP8_ACT_SCALE
        return self.q(F.silu(y))

    def conv(self, x, name, stride=1, pad=1, fp8=False):
        w = self.w[name + ".w"]
        if fp8:
            sw = w.abs().amax(dim=(1, 2, 3), keepdim=True) / 448.0
            sw = torch.where(sw > 0, sw, torch.ones_like(sw))
            w = _e4m3(w / sw) * sw
        y = F.conv2d(x, w, self.w[name + ".b"], stride=stride, padding=pad)
        return self.q(y)

    def resblock(self, x, prefix, level, film):
        f8 = self.fp8 and x.shape[1] >= 128
        h = self.conv(self.gn_film_silu(x, prefix + ".gn1", level, film), prefix + ".conv1", fp8=f8)
        self.keep(prefix + ".h", h)
        y = self.conv(self.gn_film_silu(h, prefix + ".gn2", level, film), prefix + ".conv2", fp8=f8)
        out = self.q(y + x)
        self.keep(prefix, out)
        return out

    def forward(self, rgb_u8, scores):
        x_in = torch.from_numpy(np.ascontiguousarray(rgb_u8)).permute(0, 3, 1, 2).float()  # raw 0..255
        cond = torch.from_numpy(np.asarray(scores, dtype=np.float64)).float()  # the engine casts double -> float
        film = cond @ self.w["film.w"].t() + self.w["film.b"][None, :]
        x = F.conv2d(x_in, self.w["stem.w"], self.w["stem.b"], padding=1)
        x = self.q(x)
        self.keep("stem", x)
        skips = []
        for l in range(4):
            x = self.resblock(x, f"enc{l}.rb0", l, film)
            x = self.resblock(x, f"enc{l}.rb1", l, film)
            if l < 3:
                skips.append(x)
                x = self.conv(x, f"down{l}", stride=2)
                self.keep(f"down{l}", x)
        x = self.resblock(x, "mid.rb0", 3, film)
        x = self.resblock(x, "mid.rb1", 3, film)
        for l in (2, 1, 0):
            x = F.interpolate(x, scale_factor=2, mode="nearest")
            x = self.conv(x, f"up{l}")
            self.keep(f"up{l}", x)
            x = torch.cat([x, skips[l]], dim=1)
            x = self.conv(x, f"fuse{l}", pad=0)
            self.keep(f"fuse{l}", x)
            x = self.resblock(x, f"dec{l}.rb0", l, film)
            x = self.resblock(x, f"dec{l}.rb1", l, film)
        y = self.gn_film_silu(x, "head.gn", 0, film)
        head = F.conv2d(y, self.w["head.w"], self.w["head.b"], padding=1)
        if self.cap is not None:
            self.cap["head"] = head.permute(0, 2, 3, 1).contiguous().numpy().copy()
        out = torch.clamp(x_in + head, 0.0, 255.0)
        out = torch.floor(out + 0.5)
        return out.permute(0, 2, 3, 1).contiguous().to(torch.uint8).numpy()


def restore(rgb_u8, scores, weights, emulate_bf16=False, capture=None, threads=None, emulate_fp8=False):
    """rgb_u8 [N,H,W,3] uint8, scores [N,7] -> restored [N,H,W,3] uint8."""
    rgb_u8 = np.asarray(rgb_u8)
    if rgb_u8.ndim == 3:
        rgb_u8 = rgb_u8[None]
    if threads:
        torch.set_num_threads(int(threads))
    with torch.no_grad():
        return _Net(weights, emulate_bf16, capture, emulate_fp8).forward(rgb_u8, np.asarray(scores).reshape(rgb_u8.shape[0], 7))
