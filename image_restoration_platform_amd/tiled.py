"""cfg 4 of BASELINE.json: one large image (the reference caps uploads at 2048 px: imagePreprocess.js:4) restored as row
strips -- one strip per GPU of the node, or all strips on one GPU as "virtual ranks" (Engine.restore_tiled_tensor).

restore_strip() is one rank's side of the multi-GPU flow: the engine runs the layer program one op at a time
(ire_strips_run_op); after every op whose output a 3x3 convolution reads, the boundary rows travel to the neighbouring
ranks (sharding.exchange_halos: point-to-point over the direct xGMI links); after every op that wrote GroupNorm partial
statistics the per-tile partials of all ranks are all-gathered (sharding.allgather_parts).  Nothing else is exchanged;
the result is bit-identical to the untiled run.
"""
import json

import numpy as np


def split_rows(img_u8, rank, world):
    """rows of `img_u8` [H,W,3] (torch, any device) rank owns, with one halo row above and below (zeros outside the image: never read)."""
    import torch
    h = img_u8.shape[0]
    hr = h // world
    y0 = rank * hr
    out = torch.zeros((hr + 2,) + tuple(img_u8.shape[1:]), dtype=img_u8.dtype, device=img_u8.device)
    lo, hi = max(0, y0 - 1), min(h, y0 + hr + 1)
    out[lo - (y0 - 1):lo - (y0 - 1) + (hi - lo)] = img_u8[lo:hi]
    return out


def restore_strip(sess, rows_with_halo, scores, sharding, stream=None, grouped=True, counter=None):
    """This rank's strip through the network.  sess: engine.StripSession (1 strip per rank); returns the strip's restored rows.
    grouped (default): ONE exchange launch per op -- its halo rows and its slice of the GroupNorm partials in one group of
    point-to-point operations (sharding.exchange_step); grouped=False: round 2's two launches per op (halo exchange, then an
    all-gather), kept for A/B.  counter: a dict whose "exchanges" entry is incremented per exchange launch."""
    sess.set_input(rows_with_halo, scores, stream)
    n = 0
    for k in range(sess.num_ops):
        info = sess.run_op(k, stream)
        hb, loc = int(info.halo_bytes), int(info.stats_local_bytes)
        if grouped:
            if hb:
                sess.pack_halo(k, stream)
            n += sharding.exchange_step(sess.send_up, sess.send_down, sess.recv_up, sess.recv_down, hb,
                                        sess.stats, int(info.stats_offset_bytes), loc, int(info.stats_total_bytes))
            if hb:
                sess.unpack_halo(k, stream)
            continue
        if hb:
            sess.pack_halo(k, stream)
            sharding.exchange_halos(sess.send_up, sess.send_down, sess.recv_up, sess.recv_down, hb)
            sess.unpack_halo(k, stream)
            n += 1
        if loc:
            sharding.allgather_parts(sess.stats, int(info.stats_offset_bytes), loc, int(info.stats_total_bytes))
            n += 1
    if counter is not None:
        counter["exchanges"] = counter.get("exchanges", 0) + n
    return sess.get_output(stream)


def restore_strips_pipelined(sessions, rows_with_halo, scores, sharding, streams=None, counter=None):
    """Several images in flight on this rank, one StripSession (and, on a GPU, one HIP stream) each, advanced op by op in turn:
    while image A's grouped exchange for op k is on the wire, image B's op k computes -- the exchange latency of a strip-parallel
    job stream is hidden by the other job, where no split of ONE image's op can hide it (a GroupNorm'd convolution, 32 of the 43,
    cannot start a single tile before the partials of EVERY rank have arrived: not just its boundary rows wait for the exchange).
    Under RCCL the P2P operations are stream-ordered, so issuing A's exchange on A's stream and B's op on B's stream is all it
    takes; the engine serialises its own calls (Engine::enter / leave) but not the collectives between them.  Every image's
    result is the single-image flow's, bit for bit (the per-image arithmetic is untouched).  Returns the list of restored strips."""
    import contextlib
    import torch
    n_img = len(sessions)
    streams = streams or [None] * n_img
    for sess, rows, sc, st in zip(sessions, rows_with_halo, scores, streams):
        sess.set_input(rows, sc, st)
    n = 0
    for k in range(sessions[0].num_ops):
        for sess, st in zip(sessions, streams):
            info = sess.run_op(k, st)
            hb, loc = int(info.halo_bytes), int(info.stats_local_bytes)
            if hb:
                sess.pack_halo(k, st)
            with (torch.cuda.stream(st) if st is not None else contextlib.nullcontext()):      # the exchange rides the image's own stream
                n += sharding.exchange_step(sess.send_up, sess.send_down, sess.recv_up, sess.recv_down, hb,
                                            sess.stats, int(info.stats_offset_bytes), loc, int(info.stats_total_bytes))
            if hb:
                sess.unpack_halo(k, st)
    if counter is not None:
        counter["exchanges"] = counter.get("exchanges", 0) + n
    return [sess.get_output(st) for sess, st in zip(sessions, streams)]


def bench(ctx, eng, line, mfma_peak):
    """bench.py --workload tiled.  N = 1: one size x size image per step as --strips virtual ranks on the GPU.  N > 1: one strip
    per rank (N must divide the height into strips of a multiple of 128 rows), halo rows and partials over RCCL."""
    from . import synth, weights
    a, torch = ctx.args, ctx.torch
    S = a.size
    img = torch.from_numpy(synth.batch(1, S, S)[0]).to(ctx.dev)
    jp = torch.ones(1, dtype=torch.uint8, device=ctx.dev)
    scores, _ = eng.classify_tensor(img[None], jp)
    f3, f1 = weights.conv_flops(S, S)
    if ctx.world == 1:
        out = torch.empty_like(img)

        def step(i):
            eng.restore_tiled_tensor(img, a.strips, out_u8=out, scores=scores[0])
        par = "%d row strips as virtual ranks on one GPU: per-level halo rows and GroupNorm partials by in-device copies" % a.strips
    else:
        # TWO images in flight per rank (a job stream): image B's op computes while image A's grouped exchange is on the wire
        sessions = [eng.open_strips(S, S, ctx.world, ctx.rank, 1) for _ in range(2)]
        streams = [torch.cuda.Stream(device=ctx.dev) for _ in range(2)]
        rows = split_rows(img, ctx.rank, ctx.world).contiguous()

        def step(i):
            restore_strips_pipelined(sessions, [rows, rows], [scores[0], scores[0]], ctx.sharding, streams)
            for st in streams:
                torch.cuda.current_stream().wait_stream(st)
        par = ("%d row strips, one per GPU, 2 images in flight: one grouped P2P exchange per op (halo rows + GroupNorm partials) on each image's own "
               "stream, overlapped with the other image's compute (RCCL)" % ctx.world)

    def start_profile():
        eng.profile_reset()
        eng.profile_enable(2)

    dt = ctx.timed(step, before=start_profile)
    eng.profile_enable(0)
    c3 = eng.profile_query("conv3x3")
    if ctx.rank == 0:
        ach = c3["flops"] / (c3["ms"] * 1e-3) / 1e12 if c3["ms"] > 0 else 0.0
        print(json.dumps(line(ctx, "restored images/sec @%dx%d tiled (%s)" % (S, S, a.precision), (2 if ctx.world > 1 else 1) * a.steps / dt, dt, a.precision, {
            "workload": "cfg4: one %dx%d image, RestoreNet-v0 in row strips with per-level halo exchange, %.1f GFLOP/image" % (S, S, (f3 + f1) / 1e9),
            "parallelism": par}, {
            "scaling": "strong",
            "roofline": {"kernel": "conv3x3 family on rank 0 (its strips only)", "bound": "mfma", "achieved": ach, "peak": mfma_peak, "unit": "TFLOP/s",
                         "frac": ach / mfma_peak, "traffic": None, "launches": c3["launches"]}})))
