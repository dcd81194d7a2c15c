"""The PyTorch-ROCm extension host (csrc/torch_ext.cpp -> lib/_ire_torch.so; torch_host.TorchEngine): builds, loads, fails
loudly without a GPU, and on the GPU returns exactly what the ctypes host returns (both sit on the same C ABI)."""
import os

import numpy as np
import pytest

from image_restoration_platform_amd import synth, torch_host


def test_extension_is_built_and_loads_and_fails_loudly_without_gpu():
    import torch
    assert os.path.exists(torch_host.EXT_PATH), "run __graft_entry__.build()"
    ext = torch_host.load_extension()
    for name in ("init", "shutdown", "classify", "restore", "fuse", "restore_tiled", "fuse_batch", "preprocess", "encode_png_base64"):
        assert hasattr(ext, name)
    if not torch.cuda.is_available():
        from image_restoration_platform_amd.engine import EngineError
        with pytest.raises(EngineError) as e:
            torch_host.TorchEngine(weights_path=None)
        assert e.value.status == 3 and "service unavailable" in e.value.message


@pytest.mark.gpu
def test_extension_matches_the_ctypes_host(engine):
    import torch
    from image_restoration_platform_amd.engine import EngineError
    te = torch_host.TorchEngine(device_index=0, max_batch=8)
    try:
        x = torch.from_numpy(synth.batch(3, 96, 128, start=60)).cuda()
        jp = torch.tensor([1, 0, 1], dtype=torch.uint8, device="cuda")
        s1, l1 = te.classify(x, jp)
        s0, l0 = engine.classify_tensor(x, jp)
        torch.cuda.synchronize()
        assert torch.equal(s1.view(torch.int64), s0.view(torch.int64)) and torch.equal(l1, l0)
        a = te.restore(x, None, jp)
        b = engine.restore_tensor(x, is_jpeg_u8=jp)
        c = te.restore(x, s1, None)
        torch.cuda.synchronize()
        assert torch.equal(a, b) and torch.equal(c, b)
        v = torch.from_numpy(np.ascontiguousarray(synth.fusion_views(64, 64))).cuda()
        f1, sh1 = te.fuse(v, 0.2)
        f0, sh0 = engine.fuse_tensor(v, 0.2)
        torch.cuda.synchronize()
        assert torch.equal(f1, f0) and torch.equal(sh1, sh0)
        img = torch.from_numpy(synth.batch(1, 256, 128, start=61)[0]).cuda()
        t1 = te.restore_tiled(img, 2)
        w1 = te.restore(img[None])[0]
        torch.cuda.synchronize()
        assert torch.equal(t1, w1)
        # the three entry points added in round 4: batched fusion, the preprocess step, the device-side PNG + base64 text
        vb = torch.from_numpy(np.ascontiguousarray(np.stack([synth.fusion_views(64, 72, seed=5 + i) for i in range(3)]))).cuda()
        fb1, sb1 = te.fuse_batch(vb, [0.1, -1.0, 0.3])
        fb0, sb0 = engine.fuse_batch_tensor(vb, [0.1, -1.0, 0.3])
        torch.cuda.synchronize()
        assert torch.equal(fb1, fb0) and torch.equal(sb1, sb0)
        big = torch.from_numpy(synth.batch(1, 120, 200, start=62)[0]).cuda()
        p1 = te.preprocess(big, orientation=6, max_dim=96)
        p0 = engine.preprocess_tensor(big, orientation=6, max_dim=96)
        torch.cuda.synchronize()
        assert p1.shape == p0.shape and torch.equal(p1, p0)
        e1 = te.encode_png_base64(x)
        e0 = engine.encode_png_base64_tensor(x)
        torch.cuda.synchronize()
        assert torch.equal(e1, e0)
        import base64, io
        from PIL import Image
        png = base64.b64decode(bytes(e1[1].cpu().numpy()))
        assert np.array_equal(np.asarray(Image.open(io.BytesIO(png)).convert("RGB")), x[1].cpu().numpy())
        with torch.cuda.stream(torch.cuda.Stream()):            # the extension follows the CURRENT torch stream
            d = te.restore(x, None, jp)
        torch.cuda.synchronize()
        assert torch.equal(d, b)
        with pytest.raises(EngineError) as e:
            te.restore(torch.zeros((1, 60, 64, 3), dtype=torch.uint8, device="cuda"))
        assert e.value.status == 1 and "invalid" in e.value.message
    finally:
        te.close()
