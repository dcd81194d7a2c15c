"""N > 1 path of bench.py / the worker: per-image sharding with NO data-path collective; only the
timing barrier and the max-over-ranks reduction use torch.distributed.  Exercised with gloo on CPU
(world_size 2) with a stand-in for the GPU step, since the container has no GPU."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from image_restoration_platform_amd import sharding  # noqa: E402


def test_shard_plan_is_a_partition():
    for n in (1, 5, 8, 17, 64):
        for world in (1, 2, 3, 8):
            plan = [sharding.shard_range(n, r, world) for r in range(world)]
            flat = [i for a, b in plan for i in range(a, b)]
            assert flat == list(range(n))
            sizes = [b - a for a, b in plan]
            assert max(sizes) - min(sizes) <= 1
    assert sharding.fusion_groups(8) == [[0, 1, 2], [3, 4, 5]]       # 2 groups of 3 in flight, 2 GPUs left
    assert sharding.fusion_groups(3) == [[0, 1, 2]] and sharding.fusion_groups(2) == []


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = sharding.shard_range(11, rank, world)
        # stand-in for "restore my shard": a per-image checksum; results gathered like the worker would
        local = torch.tensor([float(i * i) for i in range(lo, hi)])
        dt = sharding.timed_region(lambda: None, steps=3, sync=lambda: None, slow=0.01 * (rank + 1))
        total = torch.tensor([float(local.sum())])
        dist.all_reduce(total)                                   # test-only check, not on the job path
        # fusion gather: ranks 0..1 hold a view each, rank 0 fuses
        view = torch.full((4, 4, 3), rank + 1, dtype=torch.uint8)
        views = sharding.gather_views(view, dst=0, group_ranks=[0, 1])
        q.put((rank, lo, hi, dt, float(total.item()), None if views is None else [int(v[0, 0, 0]) for v in views]))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharding_and_timing():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    (r0, lo0, hi0, dt0, tot0, v0), (r1, lo1, hi1, dt1, tot1, v1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 6, 6, 11)
    assert dt0 == pytest.approx(dt1) and dt0 >= 3 * 0.02        # MAX over ranks: the slow rank sets the time
    assert tot0 == tot1 == float(sum(i * i for i in range(11)))
    assert v0 == [1, 2] and v1 is None
