#!/usr/bin/env python3
"""bench.py -- restored images/sec for the queue-worker hot path (classify + RestoreNet-v0).

Workload (BASELINE.json metric): 1024x1024 RGB, batch 8 per GPU, synthetic,
inputs resident in HBM when the timed region starts; a "step" = one batch through
ire_restore_device (fused classifier scan + 43-conv U-Net + u8 store).  N > 1: one process per
GPU, images sharded per rank (sharding.shard_range) with NO data-path collective (jobs are
independent: restorator.js:198-211); only the timing barrier / max (sharding.timed_region) uses
torch.distributed (RCCL).

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself: N fresh child
processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set), created before this
process imports torch or touches a GPU; the parent only waits and relays rank 0's JSON line.  Under
torchrun (WORLD_SIZE set) it is a rank.

Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` (the 3x3 conv family,
HIP-event timed inside the engine on the launching stream) and `cpu_baseline` (the CPU oracle --
test infrastructure -- timed on this box's host cores, N=1 only).

Other workloads of the same path (--workload): `fusion` = cfg 3 (3 views restored -- one per rank of a
fusion group when N >= 3 -- gathered point-to-point, aligned and blended); `classify` = the 7-score
scan alone; `tiled` = cfg 4 (one 2048x2048 image restored in row strips with per-layer halo exchange
and a per-GroupNorm gather of partial statistics; --precision fp8 for the C >= 128 convs).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md chip table
MFMA_FP8_PEAK_TFLOPS = 5000.0   # dense fp8 (block-scaled MFMA), same table
HBM_PEAK_GBS = 8000.0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=None, help="image side (default 1024; 2048 for --workload tiled)")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("IRE_STREAMS", "1")),
                    help="lanes (HIP streams) per engine; 2 is ~1 %% faster (r02: 866 -> 873 img/s) but overlapping kernels skew the per-kernel event timing the roofline is computed from")
    ap.add_argument("--workload", choices=["restore", "fusion", "classify", "tiled"], default="restore",
                    help="restore = the BASELINE metric (default); fusion = cfg 3; classify = the 7-score scan alone; tiled = cfg 4")
    ap.add_argument("--precision", choices=["bf16", "fp8"], default="bf16", help="fp8: OCP e4m3 operands for the C >= 128 ResBlock convs (cfg 4)")
    ap.add_argument("--strips", type=int, default=8, help="tiled: row strips per image on ONE GPU (virtual ranks); with N > 1 ranks each rank owns one strip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-path", action="store_true", help="skip the host-buffer (PCIe-inclusive) sub-record measured after the timed region")
    ap.add_argument("--no-secondary", action="store_true", help="skip the BASELINE cfg 1 (512x512 bs 8) sub-record measured after the timed region")
    ap.add_argument("--no-profile", action="store_true", help="skip the in-engine HIP-event kernel timing")
    ap.add_argument("--profile-all", action="store_true", help="time every kernel family (more events, ~5 %% slower)")
    ap.add_argument("--profile-every", type=int, default=4, help="time the dominant family's launches on every N-th timed step (an event record is a packet "
                    "between two kernels: on every step it costs 1.35 %% of the step); 1 = every step")
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)   # tests only: gloo + a sleep instead of the GPU step
    args = ap.parse_args(argv)
    if args.size is None:
        args.size = 2048 if args.workload == "tiled" else 1024
    return args


# ------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks (fresh processes, before anything here touches a GPU)
# ------------------------------------------------------------------------------------------------------
def launch_ranks(n, argv, timeout=None):
    """Start n rank processes of this script and relay rank 0's stdout.  Returns the worst exit code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate(timeout=timeout)
    rc = procs[0].returncode
    for p in procs[1:]:
        try:
            p.wait(timeout=120)
        except subprocess.TimeoutExpired:
            p.kill()          # the exact child we started
            p.wait()
        rc = rc or p.returncode
    sys.stdout.write(out0)
    sys.stdout.flush()
    return rc


def cpu_baseline(size, budget_s=25.0):
    """Oracle (CPU restatement, kind='port') on a bounded sample (about 30 s in all): the bench workload's size with the box's
    CPU share, then SURVEY.md 8(d)'s other rows -- 512x512 (cfg 1) and 256x256 (cfg 0, the reference's own CPU-runnable case)
    with all threads and with ONE thread."""
    import numpy as np
    import torch
    from image_restoration_platform_amd import synth, weights
    from oracle import classifier as oc
    from oracle import restorenet as onet
    nproc = os.cpu_count() or 1
    cores = min(nproc, 16)   # a 1-GPU box's CPU share; more threads only oversubscribe
    w = weights.generate(0)

    def run(sz, n, threads):
        torch.set_num_threads(threads)
        img = synth.batch(n, sz, sz)
        t0 = time.perf_counter()
        sc = np.stack([oc.classify(img[i], True)[0] for i in range(n)])
        t1 = time.perf_counter()
        for i in range(n):          # one image at a time, as the reference's worker would (restorator.js:198-211)
            onet.restore(img[i:i + 1], sc[i:i + 1], w)
        t2 = time.perf_counter()
        return n / (t2 - t0), 1e3 * (t1 - t0) / n, (t2 - t1) / n

    n = 3 if size <= 1024 else 1
    v, cls_ms, net_s = run(size, n, cores)
    rows = {"%dx%d, %d threads" % (size, size, cores): round(v, 4)}
    for sz, nn, th in ((512, 2, cores), (256, 2, cores), (512, 1, 1), (256, 1, 1)):
        if sz == size and th == cores:
            continue
        rows["%dx%d, %d thread%s" % (sz, sz, th, "" if th == 1 else "s")] = round(run(sz, nn, th)[0], 4)
    torch.set_num_threads(cores)
    return {"value": v, "unit": "images/sec", "cores": cores, "nproc": nproc, "kind": "port",
            "sample": f"{n} images {size}x{size}: C classifier oracle (1 thread) {cls_ms:.0f} ms/image + "
                      f"PyTorch-CPU fp32 RestoreNet oracle ({cores} threads) {net_s:.1f} s/image",
            "images_per_sec_by_config": rows}


def host_path(eng, size, batch, jobs=64):
    """PCIe- and copy-inclusive rate of the deployed path, measured OUTSIDE the timed region (never the bench `value`):
    `jobs` single-image jobs through ire_submit / ire_poll (the batcher both service seams use) from one thread with 8 jobs in
    flight (the reference keeps 3 per restoreBatch, 5 per worker: restorator.js:13-14, design.md:851), and ire_restore on host
    batches.  The Node seams' rate is measured by tools/host_path_rate.py (needs a second engine in a node process)."""
    import collections
    from image_restoration_platform_amd import synth
    x = synth.batch(batch, size, size)
    for _ in range(2):
        eng.restore(x)
    t0 = time.perf_counter()
    for _ in range(4):
        eng.restore(x)
    host_batch = 4 * batch / (time.perf_counter() - t0)
    for j in [eng.submit(x[i % batch]) for i in range(2 * batch)]:
        eng.poll(j)
    out = {}
    for inflight in (8, 16):
        b0 = eng.stats()["batches"]
        dt = closed_loop(eng, x, jobs, inflight)
        out["submit_poll_%d_in_flight" % inflight] = {"images_per_sec": jobs / dt, "engine_batches": eng.stats()["batches"] - b0, "jobs": jobs}
    out["ire_restore_host_batches"] = {"images_per_sec": host_batch, "batch": batch}
    cpus, node = eng.affinity()
    out["service_thread_affinity"] = {"cpulist": cpus, "numa_node": node}
    return out


def closed_loop(eng, x, jobs, inflight):
    """`jobs` single-image jobs from ONE thread, `inflight` outstanding: poll the oldest, submit the next.  Seconds."""
    import collections
    t0 = time.perf_counter()
    q = collections.deque()
    for i in range(jobs):
        if len(q) == inflight:
            eng.poll(q.popleft())
        q.append(eng.submit(x[i % len(x)]))
    while q:
        eng.poll(q.popleft())
    return time.perf_counter() - t0


def host_path_ranks(ctx, eng, size, batch, jobs=64, inflight=16):
    """N > 1: EVERY rank feeds its GPU from host memory at once -- one submit thread per GPU, 16 single-image jobs in flight, the
    batcher's pinned staging and its service threads on the GPU's NUMA node (csrc/affinity.hpp) -- between two barriers; the time
    is the MAX over ranks like the step time (the 8-GPU target is about host feeding; restorator.js:198-211).
    Returns (on rank 0) the whole-host rate and each rank's own rate and CPU plan."""
    from image_restoration_platform_amd import synth
    x = synth.batch(batch, size, size, start=ctx.rank * batch)
    for j in [eng.submit(x[i % batch]) for i in range(2 * batch)]:
        eng.poll(j)
    ctx.sync()
    if ctx.world > 1:
        ctx.dist.barrier()
    dt = closed_loop(eng, x, jobs, inflight)
    cpus, node = eng.affinity()
    mine = {"rank": ctx.rank, "images_per_sec": jobs / dt, "seconds": dt, "cpulist": cpus, "numa_node": node}
    rows = [mine]
    if ctx.world > 1:
        rows = [None] * ctx.world
        ctx.dist.all_gather_object(rows, mine)
    if ctx.rank != 0:
        return None
    worst = max(r["seconds"] for r in rows)
    return {"jobs_per_rank": jobs, "in_flight_per_rank": inflight, "images_per_sec": ctx.world * jobs / worst, "seconds_max_over_ranks": worst,
            "per_rank": rows, "pcie_bytes_per_image": 2 * 3 * size * size,
            "note": "host buffers -> pinned staging -> H2D -> restore -> D2H -> host buffers on every rank at once; never the bench `value`"}


def per_level_roofline(groups, steps, counters=None):
    """roofline.per_group / per_level: each layer group of the step against BOTH rooflines ("reporting the
    fraction per level").  For a group: t_mfma = executed flops / MFMA peak, t_hbm = algorithmic bytes / HBM peak, the larger
    one is the group's roofline time and names what binds it; frac_of_bound = that time / the measured time.  `counters`
    (optional): per-group HBM bytes per launch from the committed rocprofv3 --pmc passes."""
    rows, lv = [], {}
    tot_roof = tot_ms = 0.0
    for g in groups:
        if g["launches"] == 0 or g["ms"] <= 0:
            continue
        t = g["ms"] * 1e-3
        t_mfma = g["flops_executed"] / (MFMA_BF16_PEAK_TFLOPS * 1e12)
        t_hbm = g["bytes"] / (HBM_PEAK_GBS * 1e9)
        bound = "mfma" if t_mfma >= t_hbm else "hbm"
        cb = (counters or {}).get(g["group"])
        row = {"group": g["group"], "kernel": g["kernel"], "cin": g["cin"], "cout": g["cout"], "launches_per_step": g["launches"] / steps,
               "us_per_launch": 1e3 * g["ms"] / g["launches"], "ms_per_step": g["ms"] / steps,
               "gflop_algorithmic_per_launch": g["flops"] / g["launches"] / 1e9, "gflop_executed_per_launch": g["flops_executed"] / g["launches"] / 1e9,
               "mb_algorithmic_per_launch": g["bytes"] / g["launches"] / 1e6,
               "mfma_frac_algorithmic": g["flops"] / t / 1e12 / MFMA_BF16_PEAK_TFLOPS, "mfma_frac_executed": t_mfma / t,
               "hbm_frac_algorithmic": t_hbm / t, "bound": bound, "frac_of_bound": max(t_mfma, t_hbm) / t}
        if cb:
            row["mb_counter_per_launch"] = cb["hbm_bytes_per_launch"] / 1e6
            row["hbm_frac_counter"] = cb["hbm_bytes_per_launch"] * g["launches"] / t / 1e9 / HBM_PEAK_GBS
        rows.append(row)
        key = g["group"].split(".")[0] if g["group"].startswith("L") else "".join(c for c in g["group"] if not c.isdigit())
        a = lv.setdefault(key, {"ms": 0.0, "t_mfma": 0.0, "t_hbm": 0.0, "flops": 0.0, "t_roof": 0.0})
        a["ms"] += g["ms"]; a["t_mfma"] += t_mfma; a["t_hbm"] += t_hbm; a["flops"] += g["flops"]; a["t_roof"] += max(t_mfma, t_hbm)
        if g["group"].startswith(("L", "up", "down")):
            tot_roof += max(t_mfma, t_hbm); tot_ms += g["ms"]
    per_level = {k: {"ms_per_step": v["ms"] / steps, "mfma_frac_algorithmic": v["flops"] / (v["ms"] * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                     "mfma_frac_executed": v["t_mfma"] / (v["ms"] * 1e-3), "hbm_frac_algorithmic": v["t_hbm"] / (v["ms"] * 1e-3),
                     "bound": "mfma" if v["t_mfma"] >= v["t_hbm"] else "hbm", "frac_of_bound": v["t_roof"] / (v["ms"] * 1e-3)}
                 for k, v in sorted(lv.items())}
    return {"per_group": rows, "per_level": per_level,
            "frac_of_per_layer_roofline": (tot_roof / (tot_ms * 1e-3)) if tot_ms > 0 else None,
            "per_layer_roofline_note": "sum over the conv3x3 layer groups of max(executed flops / 2.5 PFLOP/s, algorithmic bytes / 8 TB/s) divided by their measured time: "
                                       "1.0 = every layer on whichever table roofline binds it"}


class Ctx:
    """Rank context: distributed state + the timing contract (sharding.timed_region)."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        from image_restoration_platform_amd import sharding
        self.args, self.torch, self.dist, self.sharding = args, torch, dist, sharding
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != max(1, args.gpus) and self.rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world}: reporting n_gpus={self.world}", file=sys.stderr)
        if args.stub:
            self.dev = torch.device("cpu")
            backend = "gloo"
        else:
            if not torch.cuda.is_available():
                raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
            torch.cuda.set_device(self.local_rank)
            self.dev = torch.device("cuda", self.local_rank)
            backend = "nccl"
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            kw = {} if args.stub else {"device_id": self.dev}
            dist.init_process_group(backend, rank=self.rank, world_size=self.world, **kw)

    def sync(self):
        if not self.args.stub:
            self.torch.cuda.synchronize()

    def timed(self, step, before=None):
        """W untimed warm-up steps, then exactly K steps between barrier + synchronize on both sides, MAX over ranks."""
        for i in range(self.args.warmup):
            step(-1 - i)
        self.sync()
        if before is not None:
            before()
        return self.sharding.timed_region(step, self.args.steps, self.sync)

    def close(self):
        if self.world > 1:
            self.dist.destroy_process_group()


def line(ctx, metric, value, dt, dtype, config, extra=None):
    a = ctx.args
    res = {"metric": metric, "value": value, "unit": "images/sec", "n_gpus": ctx.world, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype,
           "data": "synthetic", "config": config}
    res.update(extra or {})
    return res


# ------------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------------
def run_stub(ctx):
    """Test stand-in for the GPU step (tests/test_distributed.py): the same sharding, timing and reporting code over gloo.
    IRE_STUB_SYSFS / IRE_STUB_BDFS (a fabricated sysfs tree and one PCI address per rank): every rank asks libire.so for the CPU plan
    of ITS GPU (ire_affinity_plan: host arithmetic, no device) and rank 0 reports all of them, as host_path_ranks does on a node."""
    a = ctx.args
    lo, hi = ctx.sharding.shard_range(a.batch * ctx.world, ctx.rank, ctx.world)
    dt = ctx.timed(lambda i: time.sleep(0.002 * (ctx.rank + 1)))
    plans = None
    if os.environ.get("IRE_STUB_SYSFS"):
        import ctypes
        from image_restoration_platform_amd import _lib
        lib = _lib.load()
        bdf = os.environ["IRE_STUB_BDFS"].split(",")[ctx.rank]
        buf = ctypes.create_string_buffer(1024)
        node, slot, nslots = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        assert lib.ire_affinity_plan(os.environ["IRE_STUB_SYSFS"].encode(), bdf.encode(), buf, len(buf), ctypes.byref(node), ctypes.byref(slot), ctypes.byref(nslots)) == 0
        mine = {"rank": ctx.rank, "bdf": bdf, "cpulist": buf.value.decode(), "numa_node": node.value, "slot": slot.value, "nslots": nslots.value}
        plans = [mine]
        if ctx.world > 1:
            plans = [None] * ctx.world
            ctx.dist.all_gather_object(plans, mine)
    if ctx.rank == 0:
        print(json.dumps(line(ctx, "stub images/sec", ctx.world * a.steps * a.batch / dt, dt, "u8",
                              {"workload": "stub", "shard_of_rank0": [lo, hi], "global_batch": a.batch * ctx.world},
                              {"host_path": {"per_rank": plans}} if plans else None)))


def run_restore(ctx, eng):
    import numpy as np
    from image_restoration_platform_amd import synth, weights
    a, torch = ctx.args, ctx.torch
    B, S = a.batch, a.size
    # per-rank shard of the job stream (weak scaling: B images per rank): rank r restores images [lo, hi)
    lo, hi = ctx.sharding.shard_range(B * ctx.world, ctx.rank, ctx.world)
    x = torch.from_numpy(synth.batch(hi - lo, S, S, start=lo)).to(ctx.dev)
    jpeg = torch.ones(hi - lo, dtype=torch.uint8, device=ctx.dev)
    out = torch.empty_like(x)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]    # per-step GPU time, no host sync inside

    def step(i):
        if i >= 0:
            marks[i].record()
        eng.restore_tensor(x, out, scores=None, is_jpeg_u8=jpeg)
        if i == a.steps - 1:
            marks[a.steps].record()

    def start_profile():                       # after the warm-up: the in-engine per-kernel event timing covers the timed steps only
        if not a.no_profile:
            eng.profile_reset()
            eng.profile_enable(1 if a.profile_all else 2 | (max(1, a.profile_every) << 8))

    if a.streams > 1:
        a.profile_every = 1          # (with several lanes a step is several passes of the network: every one of them is timed)
    n_prof = a.steps if a.profile_all else len(range(0, a.steps, max(1, a.profile_every)))     # timed steps whose launches carry events
    dt = ctx.timed(step, before=start_profile)
    prof = None
    if not a.no_profile:
        eng.profile_enable(0)
        prof = {f: eng.profile_query(f) for f in ("conv3x3", "conv1x1", "stem", "head", "classifier", "gn_finalize", "all")}
        report = eng.profile_report()
    hp_ranks = host_path_ranks(ctx, eng, S, B) if (ctx.world > 1 and not a.no_host_path) else None      # every rank takes part
    if ctx.rank != 0:
        return
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps))
    f3, f1 = weights.conv_flops(S, S)
    res = line(ctx, "restored images/sec @%dx%d bs=%d" % (S, S, B), ctx.world * a.steps * B / dt, dt, a.precision, {
        "workload": "%dx%d RGB u8, batch %d per GPU: fused classifier scan + RestoreNet-v0 (43 convs, %.1f GFLOP/image), "
                    "seeded random-init weights" % (S, S, B, (f3 + f1) / 1e9),
        "global_batch": B * ctx.world, "parallelism": "per-image data parallel x%d, no collective" % ctx.world, "streams": a.streams},
        {"step_ms_p50": step_ms[len(step_ms) // 2], "step_ms_p95": step_ms[min(len(step_ms) - 1, int(0.95 * len(step_ms)))]})
    if prof is not None:
        c3 = prof["conv3x3"]
        # HBM bytes per launch from PMC counters cannot be collected from inside this process; they come from the committed
        # rocprofv3 --pmc passes of this same command (FETCH_SIZE doubled per MI355X_MICROARCH.md), valid for 1024x1024 bs=8 only.
        traffic, traffic_src, tr = None, None, None
        for name in ("r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
            tj = os.path.join(ROOT, "profiles", name)
            if os.path.exists(tj) and S == 1024 and B == 8:
                with open(tj) as f:
                    tr = json.load(f)
                traffic, traffic_src = tr["hbm_bytes_per_launch"], "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; a recorded constant, not a measurement of this run)" % name
                break
        ach = c3["flops"] / (c3["ms"] * 1e-3) / 1e12 if c3["ms"] > 0 else 0.0
        groups = report          # mode 2 (default): the conv3x3 family's groups only; --profile-all adds stem / head / 1x1
        g3 = [g for g in groups if g["group"].startswith(("L", "up", "down"))]          # the conv3x3 family's layer groups
        exec3 = sum(g["flops_executed"] for g in g3)
        ach_x = exec3 / (c3["ms"] * 1e-3) / 1e12 if c3["ms"] > 0 and g3 else None
        res["roofline"] = {
            "kernel": "conv3x3 family: every 3x3 convolution kernel of the step (ResBlock convs incl. their folded GroupNorm finalize, stride-2 down, up composed with the 1x1 fuse)", "bound": "mfma",
            "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_BF16_PEAK_TFLOPS,
            "frac_note": "frac = ALGORITHMIC flops (2*9*Cin*Cout per output pixel; for `up` also the 1x1 fuse it is composed with) / time / peak; "
                         "frac_executed counts what the MFMA pipe really issues (the sub-pixel `up` form runs 4 of 9 taps + the skip half of the fuse): use it as MFMA-pipe utilisation",
            "achieved_executed": ach_x, "frac_executed": (ach_x / MFMA_BF16_PEAK_TFLOPS) if ach_x is not None else None,
            "traffic": traffic, "traffic_source": traffic_src, "launches": c3["launches"], "avg_launch_us": 1e3 * c3["ms"] / max(1, c3["launches"]),
            "algorithmic_gflop_per_launch": c3["flops"] / max(1, c3["launches"]) / 1e9,
            "hbm_algorithmic_GBs": c3["bytes"] / (c3["ms"] * 1e-3) / 1e9 if c3["ms"] > 0 else 0.0,
            "family_ms_per_step": {k: v["ms"] / n_prof for k, v in prof.items()},
            "timed_steps": n_prof, "timed_steps_note": "HIP events bracket every launch of the family on %d of the %d timed steps (every %s); "
                           "each event record is a packet between two kernels" % (n_prof, a.steps, "step" if a.profile_all or a.profile_every <= 1 else "%d-th" % a.profile_every),
        }
        res["roofline"].update(per_level_roofline(groups, n_prof, (tr or {}).get("per_group")))
        res["whole_net_mfma_frac"] = (f3 + f1) * B * a.steps / dt / 1e12 / MFMA_BF16_PEAK_TFLOPS
    if ctx.world == 1 and not a.no_host_path:
        res["host_path"] = host_path(eng, S, B)
    elif hp_ranks is not None:
        res["host_path"] = {"all_ranks_closed_loop": hp_ranks}
    if ctx.world == 1 and not a.no_secondary and S == 1024:
        res["secondary"] = {"cfg1_512": secondary_512(ctx, eng, B)}
    if ctx.world == 1 and not a.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(S)
    print(json.dumps(res))


def secondary_512(ctx, eng, B, steps=12, warmup=3):
    """BASELINE.json cfg 1 (512x512 bs 8, 1 GPU) as a sub-record of the default line, measured AFTER the timed region (~0.1 s): the
    same step at the other named shape, inputs resident in HBM, the conv3x3 family HIP-event timed on every step."""
    from image_restoration_platform_amd import synth, weights
    torch = ctx.torch
    S = 512
    x = torch.from_numpy(synth.batch(B, S, S)).to(ctx.dev)
    jpeg = torch.ones(B, dtype=torch.uint8, device=ctx.dev)
    out = torch.empty_like(x)
    for _ in range(warmup):
        eng.restore_tensor(x, out, scores=None, is_jpeg_u8=jpeg)
    torch.cuda.synchronize()
    eng.profile_reset()
    eng.profile_enable(2 | (1 << 8))
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.restore_tensor(x, out, scores=None, is_jpeg_u8=jpeg)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    eng.profile_enable(0)
    c3 = eng.profile_query("conv3x3")
    rep = per_level_roofline(eng.profile_report(), steps)
    f3, f1 = weights.conv_flops(S, S)
    ach = c3["flops"] / (c3["ms"] * 1e-3) / 1e12 if c3["ms"] > 0 else 0.0
    return {"workload": "512x512 RGB u8, batch %d, 1 GPU (BASELINE cfg 1)" % B, "images_per_sec": steps * B / dt, "ms_per_step": 1e3 * dt / steps, "steps": steps,
            "conv3x3_family": {"achieved": ach, "unit": "TFLOP/s", "frac": ach / MFMA_BF16_PEAK_TFLOPS, "ms_per_step": c3["ms"] / steps,
                               "frac_of_per_layer_roofline": rep["frac_of_per_layer_roofline"]},
            "per_level": rep["per_level"], "whole_net_mfma_frac": (f3 + f1) * B * steps / dt / 1e12 / MFMA_BF16_PEAK_TFLOPS}


def run_classify(ctx, eng):
    from image_restoration_platform_amd import synth
    a, torch = ctx.args, ctx.torch
    S, B = a.size, a.batch
    lo, hi = ctx.sharding.shard_range(B * ctx.world, ctx.rank, ctx.world)
    x = torch.from_numpy(synth.batch(hi - lo, S, S, start=lo)).to(ctx.dev)
    jp = torch.ones(hi - lo, dtype=torch.uint8, device=ctx.dev)

    def step(i):
        eng.classify_tensor(x, jp)

    def start_profile():
        eng.profile_reset()
        eng.profile_enable(1)

    dt = ctx.timed(step, before=start_profile)
    eng.profile_enable(0)
    pf = eng.profile_query("classifier")
    if ctx.rank == 0:
        unit_bytes = 3.0 * S * S
        ach = unit_bytes * B * a.steps / (pf["ms"] * 1e-3) / 1e9 if pf["ms"] > 0 else 0.0
        print(json.dumps(line(ctx, "classified images/sec @%dx%d bs=%d" % (S, S, B), ctx.world * a.steps * B / dt, dt, "u8", {
            "workload": "7-score classifier scan @%dx%d bs=%d" % (S, S, B), "parallelism": "independent units x%d, no collective" % ctx.world}, {
            "roofline": {"kernel": "classifier family (scan + finalize, summed per step)", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_step": unit_bytes * B,
                         "family_ms_per_step": pf["ms"] / a.steps}})))


def run_fusion(ctx, eng):
    """cfg 3.  N >= 3: ranks are dealt into fusion groups of 3 (sharding.fusion_groups); every rank of a group restores ITS view
    of the scene, the restored views travel to the group's first rank point-to-point (two peers, two xGMI links), which aligns
    and blends them (sharding.restore_views_and_fuse).  Ranks left over (and every rank when N < 3) run the whole 3-view job
    alone: restore the 3 views as one batch of 3, fuse.  value = fused images/sec over the whole job."""
    import numpy as np
    from image_restoration_platform_amd import synth
    a, torch, sh = ctx.args, ctx.torch, ctx.sharding
    S = a.size
    groups = sh.fusion_groups(ctx.world) if ctx.world >= 3 else []
    mine = next((g for g in groups if ctx.rank in g), None)
    views_np = synth.fusion_views(S, S, seed=7 + (groups.index(mine) if mine else 100 + ctx.rank))
    jp1 = torch.ones(1, dtype=torch.uint8, device=ctx.dev)
    jp3 = torch.ones(3, dtype=torch.uint8, device=ctx.dev)
    if mine is not None:
        view = torch.from_numpy(np.ascontiguousarray(views_np[mine.index(ctx.rank)])).to(ctx.dev)

        def step(i):
            sh.restore_views_and_fuse(
                view, mine, mine[0],
                restore=lambda v: eng.restore_tensor(v[None], scores=None, is_jpeg_u8=jp1)[0],
                fuse=lambda vs: eng.fuse_tensor(torch.stack(vs, 0), noise_score=-1.0)[0])
    else:
        views = torch.from_numpy(np.ascontiguousarray(views_np)).to(ctx.dev)
        buf = torch.empty_like(views)

        def step(i):
            eng.restore_tensor(views, buf, scores=None, is_jpeg_u8=jp3)
            eng.fuse_tensor(buf, noise_score=-1.0)

    def start_profile():
        eng.profile_reset()
        eng.profile_enable(1)

    dt = ctx.timed(step, before=start_profile)
    eng.profile_enable(0)
    pf = eng.profile_query("fusion")
    jobs = len(groups) + (ctx.world - 3 * len(groups))          # one fused image per group and per lone rank, per step
    # the batched entry (ire_fuse_batch_device): FUSE_SETS restored view sets of this shape fused by one pass of the kernel chain --
    # the fusion kernels alone, inputs resident, rank 0 only, outside the timed region above
    FUSE_SETS = 8
    batched = None
    if ctx.rank == 0:
        vb = torch.from_numpy(np.ascontiguousarray(np.stack([synth.fusion_views(S, S, seed=40 + i) for i in range(FUSE_SETS)]))).to(ctx.dev)
        ns = [0.1 + 0.05 * i for i in range(FUSE_SETS)]
        for _ in range(3):
            eng.fuse_batch_tensor(vb, ns)
        torch.cuda.synchronize()
        eng.profile_reset()
        eng.profile_enable(1)
        nb = 20
        t0 = time.perf_counter()
        for _ in range(nb):
            eng.fuse_batch_tensor(vb, ns)
        torch.cuda.synchronize()
        tb = (time.perf_counter() - t0) / nb
        eng.profile_enable(0)
        pb = eng.profile_query("fusion")
        bb = FUSE_SETS * 4.0 * 3 * S * S
        achb = bb / (pb["ms"] / nb * 1e-3) / 1e9 if pb["ms"] > 0 else 0.0
        batched = {"view_sets_per_call": FUSE_SETS, "fused_images_per_sec": FUSE_SETS / tb, "kernel_chain_ms": pb["ms"] / nb, "achieved": achb,
                   "unit": "GB/s", "frac": achb / HBM_PEAK_GBS, "algorithmic_bytes_per_call": bb}
    if ctx.rank == 0:
        unit_bytes = 4.0 * 3 * S * S                             # (k + 1) * 3 * H * W: read 3 views, write one image
        n_f = max(1, a.steps)
        ach = unit_bytes / (pf["ms"] / n_f * 1e-3) / 1e9 if pf["ms"] > 0 else 0.0
        print(json.dumps(line(ctx, "fused images/sec @3x%dx%d (3 views restored, gathered, aligned, blended)" % (S, S), jobs * a.steps / dt, dt, "bf16", {
            "workload": "cfg3: 3 views @%dx%d, RestoreNet-v0 each, Fusion-v0 align + blend" % (S, S),
            "parallelism": ("%d fusion group(s) of 3 ranks: one view per rank, point-to-point gather to the fusing rank; %d lone rank(s) run whole jobs"
                            % (len(groups), ctx.world - 3 * len(groups))) if groups else "whole 3-view job per rank, no exchange"}, {
            "roofline": {"kernel": "fusion family (luma, SAD searches, blend) on rank 0", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": unit_bytes,
                         "batched_entry": batched}})))


def run_tiled(ctx, eng):
    from image_restoration_platform_amd import tiled
    tiled.bench(ctx, eng, line, MFMA_FP8_PEAK_TFLOPS if ctx.args.precision == "fp8" else MFMA_BF16_PEAK_TFLOPS)


def run_rank(args):
    ctx = Ctx(args)
    try:
        if args.stub:
            return run_stub(ctx)
        from image_restoration_platform_amd.engine import Engine
        eng = Engine(device_index=ctx.local_rank, max_batch=max(args.batch, 3), num_streams=args.streams, precision=args.precision)
        {"restore": run_restore, "classify": run_classify, "fusion": run_fusion, "tiled": run_tiled}[args.workload](ctx, eng)
    finally:
        ctx.close()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, argv))       # nothing in this process has touched a GPU (torch is not even imported)
    run_rank(args)


if __name__ == "__main__":
    main()
