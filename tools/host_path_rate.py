"""PCIe-inclusive rate of the host-pointer entry points: ire_restore on numpy batches and the async
batcher (ire_submit / ire_poll) with 16 jobs in flight.  Not the bench metric (that one is HBM-resident)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from image_restoration_platform_amd import synth
from image_restoration_platform_amd.engine import Engine

S, B = 1024, 8
eng = Engine(max_batch=B)
x = synth.batch(B, S, S)
for _ in range(3):
    eng.restore(x)
t0 = time.perf_counter()
N = 10
for _ in range(N):
    eng.restore(x)
dt = time.perf_counter() - t0
print(f"ire_restore (host numpy, bs {B} @{S}^2): {N * B / dt:.1f} img/s, {1e3 * dt / N:.2f} ms per batch")
jobs = [eng.submit(x[i % B]) for i in range(16)]
for j in jobs:
    eng.poll(j)
t0 = time.perf_counter()
jobs = [eng.submit(x[i % B]) for i in range(64)]
for j in jobs:
    eng.poll(j)
dt = time.perf_counter() - t0
print(f"ire_submit/ire_poll (64 single-image jobs, coalesced): {64 / dt:.1f} img/s")
