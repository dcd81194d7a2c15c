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


def test_bench_starts_its_own_ranks_and_reports_n_gpus(tmp_path):
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment must start 2 ranks itself (fresh processes) and rank 0 must
    print ONE JSON line with n_gpus == 2.  --stub swaps the GPU step for a sleep and RCCL for gloo; the launch, sharding
    (shard_range), timing (timed_region: barrier, K steps, barrier, MAX over ranks) and reporting code are bench.py's own."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--stub", "--batch", "8"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]      # (gloo itself prints a "[Gloo] Rank 0 is connected" line)
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 4 and j["warmup"] == 1 and j["scaling"] == "weak"
    assert j["config"]["shard_of_rank0"] == [0, 8] and j["config"]["global_batch"] == 16
    # MAX over ranks: rank 1 sleeps 4 ms per step, rank 0 2 ms
    assert j["ms_per_step"] >= 4.0 and j["value"] == pytest.approx(2 * 4 * 8 / (j["ms_per_step"] * 4 / 1e3))
    # as a rank under a launcher (WORLD_SIZE set) it does not spawn: a world of 1 reports n_gpus 1
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "0", "--stub"],
                        capture_output=True, text=True, timeout=300, env=env)
    assert r1.returncode == 0 and json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][0])["n_gpus"] == 1


def test_bench_ranks_report_disjoint_cpu_plans(tmp_path):
    """8-rank host feeding (restorator.js:198-211): every rank's service threads stay on ITS GPU's NUMA node,
    on that GPU's share of the node.  `bench.py --gpus 4 --stub` over a fabricated two-socket sysfs: each rank asks libire.so for
    the plan of its own PCI address (ire_affinity_plan), rank 0 reports all of them the way host_path_ranks does on a real node:
    the plans are pairwise disjoint, each inside its GPU's node, SMT siblings together."""
    import json
    import subprocess
    from test_batcher_native import _fake_sysfs
    gpus0, gpus1 = ["0000:05:00.0", "0000:15:00.0"], ["0000:85:00.0", "0000:95:00.0"]
    _fake_sysfs(tmp_path, {0: ("0-15,32-47", gpus0), 1: ("16-31,48-63", gpus1)})
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(IRE_STUB_SYSFS=str(tmp_path), IRE_STUB_BDFS=",".join([gpus0[0], gpus1[0], gpus0[1], gpus1[1]]))     # ranks alternate sockets
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "0", "--stub"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    rows = j["host_path"]["per_rank"]
    assert [x["rank"] for x in rows] == [0, 1, 2, 3] and [x["numa_node"] for x in rows] == [0, 1, 0, 1]

    def cpus(l):
        out = set()
        for part in l.split(","):
            a, _, b = part.partition("-")
            out.update(range(int(a), int(b or a) + 1))
        return out
    sets = [cpus(x["cpulist"]) for x in rows]
    node = [set(range(0, 16)) | set(range(32, 48)), set(range(16, 32)) | set(range(48, 64))]
    for i, s in enumerate(sets):
        assert len(s) == 16 and s <= node[rows[i]["numa_node"]]
        assert all((c + 32) in s for c in s if c < 32)                 # a core and its SMT sibling go to the same rank
        for t in sets[:i]:
            assert not (s & t)
    assert set().union(*sets) == set(range(64))


def _fusion_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import numpy as np
        from oracle import fusion as ofus          # the CPU oracle is the checker here, never on the product path
        from image_restoration_platform_amd import synth
        views = synth.fusion_views(64, 64, seed=5)
        restore = lambda v: 255 - v                 # stand-in for the per-rank RestoreNet call (any per-view map)
        fuse = lambda vs: torch.from_numpy(ofus.fuse(np.stack([x.numpy() for x in vs], 0), 0.2)[0])
        group = [0, 1, 2]
        fused = sharding.restore_views_and_fuse(torch.from_numpy(views[rank].copy()), group, 0, restore, fuse)
        single = fuse([restore(torch.from_numpy(views[i].copy())) for i in range(3)]) if rank == 0 else None
        q.put((rank, None if fused is None else bool(torch.equal(fused, single)), None if fused is None else tuple(fused.shape)))
    finally:
        dist.destroy_process_group()


def test_cfg3_restore_on_three_ranks_gather_fuse_equals_single_process():
    """cfg 3 flow (restore one view per rank -> gather_views -> fuse on dst): fused bytes == the single-process result."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_fusion_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(3))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert res[0] == (0, True, (64, 64, 3)) and res[1][1] is None and res[2][1] is None


def _strip_worker(rank, world, port, q):
    """cfg 4 on CPU tensors: a toy 3-layer strip network (3x3 box sums with zero padding at the image border, then a global
    normalisation by statistics every strip contributes to) decomposed into row strips that talk ONLY through the product's
    exchange functions; the assembled result must equal the whole-image computation exactly (integer arithmetic)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import numpy as np
        H, W, hr = 8 * world, 6, 8
        rng = np.random.default_rng(3)
        img = rng.integers(0, 50, (H, W)).astype(np.int64)

        def layer(x_with_halo):            # rows 0 and -1 are halo rows (zeros where the image ends)
            p = np.pad(x_with_halo, ((0, 0), (1, 1)))
            return sum(p[dy:dy + x_with_halo.shape[0] - 2, dx:dx + W] for dy in range(3) for dx in range(3))

        def whole(x):
            for _ in range(3):
                x = layer(np.pad(x, ((1, 1), (0, 0))))
                x = x - int(x.sum()) // x.size                     # "GroupNorm": a global statistic
            return x

        x = img[rank * hr:(rank + 1) * hr].copy()
        row_bytes = W * 8
        su, sd, ru, rd = (torch.zeros(row_bytes, dtype=torch.uint8) for _ in range(4))
        parts = torch.zeros(world * 8, dtype=torch.uint8)            # one int64 partial per strip
        for _ in range(3):
            su.copy_(torch.from_numpy(x[0].copy().view(np.uint8)))
            sd.copy_(torch.from_numpy(x[-1].copy().view(np.uint8)))
            ru.zero_(); rd.zero_()
            sharding.exchange_halos(su, sd, ru, rd, row_bytes)
            up = ru.numpy().view(np.int64) if rank > 0 else np.zeros(W, np.int64)
            dn = rd.numpy().view(np.int64) if rank + 1 < world else np.zeros(W, np.int64)
            x = layer(np.concatenate([up[None], x, dn[None]], axis=0))
            parts[rank * 8:(rank + 1) * 8] = torch.from_numpy(np.array([x.sum()], np.int64).view(np.uint8))
            sharding.allgather_parts(parts, rank * 8, 8, world * 8)
            x = x - int(parts.numpy().view(np.int64).sum()) // (H * W)
        q.put((rank, x, whole(img)[rank * hr:(rank + 1) * hr]))
    finally:
        dist.destroy_process_group()


def _strip_worker_grouped(rank, world, port, q):
    """The same toy strip network with ONE grouped exchange per layer (sharding.exchange_step): the raw layer output's boundary
    rows and the layer's partial statistic travel in one group of point-to-point operations; the consumer normalises the halo
    rows it received with the same global statistic as its own rows (what the engine does: GroupNorm is applied while staging)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import numpy as np
        H, W, hr = 8 * world, 6, 8
        rng = np.random.default_rng(3)
        img = rng.integers(0, 50, (H, W)).astype(np.int64)

        def layer(x_with_halo):
            p = np.pad(x_with_halo, ((0, 0), (1, 1)))
            return sum(p[dy:dy + x_with_halo.shape[0] - 2, dx:dx + W] for dy in range(3) for dx in range(3))

        def whole(x):
            for _ in range(3):
                x = layer(np.pad(x, ((1, 1), (0, 0))))
                x = x - int(x.sum()) // x.size
            return x

        x = img[rank * hr:(rank + 1) * hr].copy()
        row_bytes = W * 8
        su, sd, ru, rd = (torch.zeros(row_bytes, dtype=torch.uint8) for _ in range(4))
        parts = torch.zeros(world * 8, dtype=torch.uint8)
        steps = 0

        def xchg(x, with_stats):
            nonlocal steps
            su.copy_(torch.from_numpy(x[0].copy().view(np.uint8)))
            sd.copy_(torch.from_numpy(x[-1].copy().view(np.uint8)))
            ru.zero_(); rd.zero_()
            if with_stats:
                parts[rank * 8:(rank + 1) * 8] = torch.from_numpy(np.array([x.sum()], np.int64).view(np.uint8))
            steps += sharding.exchange_step(su, sd, ru, rd, row_bytes, parts, rank * 8, 8 if with_stats else 0, world * 8)
            up = ru.numpy().view(np.int64).copy() if rank > 0 else None
            dn = rd.numpy().view(np.int64).copy() if rank + 1 < world else None
            return up, dn

        up, dn = xchg(x, False)                     # the input image's boundary rows
        for i in range(3):
            zero = np.zeros(W, np.int64)
            x = layer(np.concatenate([(up if up is not None else zero)[None], x, (dn if dn is not None else zero)[None]], axis=0))
            if i < 2:
                up, dn = xchg(x, True)              # ONE step: halo rows of the raw output + its statistic
            else:
                parts[rank * 8:(rank + 1) * 8] = torch.from_numpy(np.array([x.sum()], np.int64).view(np.uint8))
                steps += sharding.exchange_step(su, sd, ru, rd, 0, parts, rank * 8, 8, world * 8)     # statistics only
            mean = int(parts.numpy().view(np.int64).sum()) // (H * W)
            x = x - mean
            if i < 2:
                up = up - mean if up is not None else None
                dn = dn - mean if dn is not None else None
        q.put((rank, x, whole(img)[rank * hr:(rank + 1) * hr], steps))
    finally:
        dist.destroy_process_group()


def test_cfg4_grouped_exchange_step_reproduces_the_whole_image():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_strip_worker_grouped, args=(r, 3, port, q)) for r in range(3)]
    for p in ps:
        p.start()
    res = [q.get(timeout=180) for _ in range(3)]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, got, want, steps in res:
        assert (got == want).all(), rank
        assert steps == 4, steps          # input rows + one grouped step per layer (the ungrouped form needs 3 + 3)


def test_cfg4_strip_exchange_functions_reproduce_the_whole_image():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_strip_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in ps:
        p.start()
    res = [q.get(timeout=180) for _ in range(3)]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, got, want in res:
        assert (got == want).all(), rank
