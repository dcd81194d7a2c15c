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


# ---- cfg 4: row strips of one large image, one strip per rank -------------------------------------------
def _staged(t):
    """gloo moves host memory: stage device tensors through the host under gloo (tests / single-GPU rehearsal); RCCL takes them as is."""
    return t.is_cuda and dist.get_backend() != "nccl"


def exchange_halos(send_up, send_down, recv_up, recv_down, nbytes, rank=None, world=None):
    """Per-level halo exchange between neighbouring strips: my first row goes to the rank above (its lower halo), my last row to
    the rank below (its upper halo); theirs arrive in recv_up / recv_down.  Point-to-point on the direct xGMI links of the two
    neighbours (batch_isend_irecv = one grouped RCCL launch), never a collective.  Tensors are flat uint8; nbytes <= their size."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    if world == 1 or nbytes == 0:
        return
    stage = _staged(send_up)
    su, sd = (send_up[:nbytes].cpu(), send_down[:nbytes].cpu()) if stage else (send_up[:nbytes], send_down[:nbytes])
    ru = torch.empty(nbytes, dtype=torch.uint8) if stage else recv_up[:nbytes]
    rd = torch.empty(nbytes, dtype=torch.uint8) if stage else recv_down[:nbytes]
    ops = []
    if rank > 0:
        ops += [dist.P2POp(dist.isend, su, rank - 1), dist.P2POp(dist.irecv, ru, rank - 1)]
    if rank + 1 < world:
        ops += [dist.P2POp(dist.isend, sd, rank + 1), dist.P2POp(dist.irecv, rd, rank + 1)]
    for q in dist.batch_isend_irecv(ops):
        q.wait()
    if stage:
        if rank > 0:
            recv_up[:nbytes].copy_(ru)
        if rank + 1 < world:
            recv_down[:nbytes].copy_(rd)


def allgather_parts(buf, offset, local, total):
    """GroupNorm partial statistics: every rank wrote `local` bytes at `offset` of the flat uint8 tensor `buf` (equal slices,
    rank order); afterwards every rank holds all `total` bytes.  An all-GATHER of the per-tile partials, not an all-reduce:
    the finalize then sums exactly the partials of the untiled run in exactly its order, so the result is bit-identical
    (and the payload is small: 64 B per tile).  In place under RCCL."""
    world = dist.get_world_size()
    if world == 1 or local == 0:
        return
    assert local * world == total and offset == dist.get_rank() * local
    if _staged(buf):
        mine = buf[offset:offset + local].cpu()
        full = torch.empty(total, dtype=torch.uint8)
        dist.all_gather_into_tensor(full, mine)
        buf[:total].copy_(full)
    else:
        dist.all_gather_into_tensor(buf[:total], buf[offset:offset + local])


def exchange_step(send_up, send_down, recv_up, recv_down, halo_bytes, stats_buf, offset, local, total, rank=None, world=None):
    """ONE grouped launch per op (cfg 4): the op's halo rows (to / from the two neighbouring strips) AND its slice of the
    GroupNorm partials (to / from every other rank) as point-to-point operations of a single batch_isend_irecv group
    (= one ncclGroupStart / End under RCCL): per-op exchange latency is paid once, not once for the halo rows and once for
    an all-gather.  The partials travel rank-to-rank on the direct xGMI links (a full mesh: every pair has its own link), in
    place: rank r's slice lands at r * local of every rank's `stats_buf`, exactly what allgather_parts produced, so the
    finalize still adds the untiled run's partials in the untiled run's order (bit-identical results).
    halo_bytes == 0: no halo rows for this op; local == 0: no statistics.  Returns the number of exchange launches (0 or 1)."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    if world == 1 or (halo_bytes == 0 and local == 0):
        return 0
    ops, post = [], []
    if halo_bytes:
        stage = _staged(send_up)
        su, sd = (send_up[:halo_bytes].cpu(), send_down[:halo_bytes].cpu()) if stage else (send_up[:halo_bytes], send_down[:halo_bytes])
        ru = torch.empty(halo_bytes, dtype=torch.uint8) if stage else recv_up[:halo_bytes]
        rd = torch.empty(halo_bytes, dtype=torch.uint8) if stage else recv_down[:halo_bytes]
        if rank > 0:
            ops += [dist.P2POp(dist.isend, su, rank - 1), dist.P2POp(dist.irecv, ru, rank - 1)]
            if stage:
                post.append((recv_up[:halo_bytes], ru))
        if rank + 1 < world:
            ops += [dist.P2POp(dist.isend, sd, rank + 1), dist.P2POp(dist.irecv, rd, rank + 1)]
            if stage:
                post.append((recv_down[:halo_bytes], rd))
    if local:
        assert local * world == total and offset == rank * local
        stage = _staged(stats_buf)
        mine = stats_buf[offset:offset + local].cpu() if stage else stats_buf[offset:offset + local]
        for r in range(world):
            if r == rank:
                continue
            dst = torch.empty(local, dtype=torch.uint8) if stage else stats_buf[r * local:(r + 1) * local]
            ops += [dist.P2POp(dist.isend, mine, r), dist.P2POp(dist.irecv, dst, r)]
            if stage:
                post.append((stats_buf[r * local:(r + 1) * local], dst))
    if not ops:
        return 0
    for q in dist.batch_isend_irecv(ops):
        q.wait()
    for dev_t, host_t in post:
        dev_t.copy_(host_t)
    return 1
