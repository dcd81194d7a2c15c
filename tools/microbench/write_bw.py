"""Pure-write / copy bandwidth of the box with ideal access patterns (torch fill / copy kernels), for pricing write-bound layers."""
import torch, time
n = 537 * 1024 * 1024
x = torch.empty(n, dtype=torch.uint8, device="cuda")
y = torch.empty(n, dtype=torch.uint8, device="cuda")
for name, f, b in (("fill", lambda: x.zero_(), n), ("copy", lambda: y.copy_(x), 2 * n)):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): f()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    print(f"{name}: {ms * 1e3:.1f} us for {b / 1e6:.0f} MB = {b / ms / 1e9:.2f} TB/s")
