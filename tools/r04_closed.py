import sys, os, time, collections
sys.path.insert(0, "/root/repo")
import numpy as np
from image_restoration_platform_amd import synth
from image_restoration_platform_amd.engine import Engine
S, B = 1024, 8
eng = Engine(max_batch=B)
x = synth.batch(B, S, S)
for _ in range(2): eng.restore(x)
jobs = [eng.submit(x[i % B]) for i in range(16)]
for j in jobs: eng.poll(j)
for inflight in (16, 8):
    print("=== inflight", inflight, file=sys.stderr, flush=True)
    b0 = eng.stats()["batches"]; t0 = time.perf_counter(); q = collections.deque()
    for i in range(64):
        if len(q) == inflight: eng.poll(q.popleft())
        q.append(eng.submit(x[i % B]))
    while q: eng.poll(q.popleft())
    dt = time.perf_counter() - t0
    print(f"closed loop {inflight}: {64/dt:.1f} img/s, {eng.stats()['batches']-b0} batches", flush=True)
