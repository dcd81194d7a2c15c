"""Multi-GPU layout of the hot path: one process per GPU, jobs sharded per image.

Single-image restoration needs NO collective (the reference already treats every image as an
independent job: restorator.js:198-211, one BullMQ job per image); the only exchange step is the
optional <=3-view fusion, where each view is restored on its own GPU and the restored views are
gathered to the fusing rank over the direct xGMI peer links (point-to-point send/recv, two peers
sending concurrently on two different links -- not a ring).  torch.distributed's "nccl" backend IS
RCCL on ROCm; tests run the same code over gloo on CPU.
"""
import time

import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous, balanced partition of n_items jobs: rank r gets [lo, hi)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def fusion_groups(world, k=3):
    """Groups of k ranks that fuse together; leftover ranks keep doing single-image work."""
    return [list(range(g * k, g * k + k)) for g in range(world // k)]


def timed_region(step, steps, sync, slow=0.0):
    """bench.py's contract: barrier + sync on both sides, K steps, MAX over ranks.  step(i) if it takes an argument."""
    import inspect
    takes_i = len(inspect.signature(step).parameters) >= 1
    if dist.is_initialized():
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i) if takes_i else step()
        if slow:
            time.sleep(slow)
    sync()
    if dist.is_initialized():
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist.is_initialized() and dist.get_world_size() > 1:
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def gather_views(view, dst, group_ranks):
    """Fusion gather: every rank in group_ranks holds one restored view [H,W,3] u8 (device tensor
    under RCCL); the views arrive at `dst` in group order over point-to-point links.  Returns the
    list of views on dst, None elsewhere."""
    rank = dist.get_rank()
    if rank not in group_ranks:
        return None
    if rank == dst:
        views, reqs = [], []
        for r in group_ranks:
            if r == dst:
                views.append(view)
            else:
                buf = torch.empty_like(view)
                reqs.append(dist.irecv(buf, src=r))   # all peers stream concurrently, one link each
                views.append(buf)
        for q in reqs:
            q.wait()
        return views
    dist.send(view.contiguous(), dst=dst)
    return None


def restore_views_and_fuse(view, group_ranks, dst, restore, fuse):
    """cfg 3 (geminiClient.js:32,49 = restoreImage with 2..3 images): every rank of the group restores
    ITS view, the restored views travel to `dst` point-to-point (gather_views), `dst` aligns and blends them.
    restore(view) -> restored view (same shape/dtype), fuse(list of views in group order) -> fused image.
    Returns the fused image on dst, None on the other ranks (and on ranks outside the group, which keep doing
    single-image work)."""
    if dist.get_rank() not in group_ranks:
        return None
    restored = restore(view)
    views = gather_views(restored, dst, group_ranks)
    if views is None:
        return None
    return fuse(views)
