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
b0 = eng.stats()["batches"]
t0 = time.perf_counter()
jobs = [eng.submit(x[i % B]) for i in range(64)]
for j in jobs:
    eng.poll(j)
dt = time.perf_counter() - t0
print(f"ire_submit/ire_poll (64 single-image jobs queued at once, coalesced): {64 / dt:.1f} img/s, {eng.stats()['batches'] - b0} engine batches")
import collections
for inflight in (3, 5, 8, 16):          # closed loop from ONE thread: poll the oldest, submit the next
    b0 = eng.stats()["batches"]
    t0 = time.perf_counter()
    q = collections.deque()
    for i in range(64):
        if len(q) == inflight:
            eng.poll(q.popleft())
        q.append(eng.submit(x[i % B]))
    while q:
        eng.poll(q.popleft())
    dt = time.perf_counter() - t0
    print(f"ire_submit/ire_poll closed loop, {inflight} in flight: {64 / dt:.1f} img/s, {eng.stats()['batches'] - b0} engine batches per 64 jobs")

# the Node seams (N-API shim -> ire_submit / ire_poll): in-flight single-image restoreImage calls, raw codec
import json, shutil, subprocess
from image_restoration_platform_amd import weights
if shutil.which("node"):
    nd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "image_restoration_platform_amd", "node", "rate_adapters.js")
    eng.close()
    for inflight in (3, 5, 8):
        r = subprocess.run(["node", nd, weights.ensure_default(0), str(S), str(inflight), "64"], capture_output=True, text=True, timeout=600)
        j = json.loads(r.stdout.strip().splitlines()[-1])
        if "fatal" in j:
            print("Node:", j)
            continue
        print(f"Node, {inflight} in flight @{S}^2: shim alone (addon.restoreAsync) {j['shim']['imagesPerSec']:.1f} img/s, {j['shim']['engineBatches']} engine batches per {j['total']} jobs; "
              f"whole seam (restoreImage: raw codec + base64 string) {j['seam']['imagesPerSec']:.1f} img/s, {j['seam']['engineBatches']} batches "
              f"(V8 base64 of one raw image: {j['jsThreadBase64MsPerJob']:.2f} ms on the JS thread = a {j['jsThreadBound']:.0f} img/s bound by itself)")
