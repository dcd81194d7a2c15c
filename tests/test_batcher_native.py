"""The batcher's state machine (csrc/batcher.hpp) under ThreadSanitizer and AddressSanitizer + UBSan on the CPU box, over a host-only
stub backend (tests/native/batcher_stress.cpp -- test infrastructure; libire.so instantiates the same template over HIP), and the
host arithmetic of the service threads' CPU plan (csrc/affinity.hpp through ire_affinity_plan).  SURVEY.md section 5 asks for the
sanitizers on the CPU build; the reference paths behind this: restorator.js:181-236 (in-flight promises), utils/retry.js:12-47
(a timed-out attempt is abandoned and resubmitted), restorator.js:198-211 (one independent job per image, 8 ranks per host)."""
import ctypes
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "batcher_stress.cpp")


def _build_and_run(tmp_path, name, flags):
    exe = str(tmp_path / name)
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer"] + flags + [SRC, "-o", exe, "-lpthread"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    out = []
    for _ in range(3):          # three runs: the interleavings differ
        r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
        out.append(r.stdout + r.stderr)
        assert r.returncode == 0, out[-1][-4000:]
        assert "batcher_stress ok" in r.stdout
    return "\n".join(out)


def test_batcher_state_machine_under_thread_sanitizer(tmp_path):
    # IRE_BATCHER_SYSCLOCK_WAITS: gcc 11's libtsan does not intercept pthread_cond_clockwait (batcher.hpp::wait_deadline)
    log = _build_and_run(tmp_path, "bs_tsan", ["-fsanitize=thread", "-DIRE_BATCHER_SYSCLOCK_WAITS"])
    assert "WARNING: ThreadSanitizer" not in log, log[-4000:]


def test_batcher_state_machine_under_address_sanitizer(tmp_path):
    log = _build_and_run(tmp_path, "bs_asan", ["-fsanitize=address,undefined"])
    assert "ERROR: AddressSanitizer" not in log and "runtime error" not in log and "LeakSanitizer" not in log, log[-4000:]


def _fake_sysfs(root, nodes):
    """nodes: {node: (cpulist, [gpu bdfs])} plus one non-GPU device per node"""
    for node, (cpulist, bdfs) in nodes.items():
        nd = root / "devices" / "system" / "node" / ("node%d" % node)
        nd.mkdir(parents=True)
        (nd / "cpulist").write_text(cpulist + "\n")
        for bdf in bdfs:
            d = root / "bus" / "pci" / "devices" / bdf
            d.mkdir(parents=True)
            (d / "numa_node").write_text("%d\n" % node)
            (d / "vendor").write_text("0x1002\n")
            (d / "class").write_text("0x120000\n")
        nic = root / "bus" / "pci" / "devices" / ("0000:%02x:00.0" % (0xf0 + node))
        nic.mkdir(parents=True)
        (nic / "numa_node").write_text("%d\n" % node)
        (nic / "vendor").write_text("0x15b3\n")
        (nic / "class").write_text("0x020000\n")


def _plan(lib, root, bdf):
    buf = ctypes.create_string_buffer(1024)
    node, slot, nslots = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    rc = lib.ire_affinity_plan(str(root).encode(), bdf.encode(), buf, len(buf), ctypes.byref(node), ctypes.byref(slot), ctypes.byref(nslots))
    assert rc == 0, lib.ire_last_error()
    cpus = set()
    for part in filter(None, buf.value.decode().split(",")):
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus, node.value, slot.value, nslots.value


def test_affinity_plan_partitions_a_two_socket_eight_gpu_host(tmp_path):
    from image_restoration_platform_amd import _lib
    lib = _lib.load()
    gpus0 = ["0000:05:00.0", "0000:15:00.0", "0000:65:00.0", "0000:75:00.0"]
    gpus1 = ["0000:85:00.0", "0000:95:00.0", "0000:e5:00.0", "0000:f5:00.0"]
    _fake_sysfs(tmp_path, {0: ("0-63,128-191", gpus0), 1: ("64-127,192-255", gpus1)})
    seen = set()
    for node, gpus in ((0, gpus0), (1, gpus1)):
        for k, bdf in enumerate(gpus):
            cpus, nd, slot, nslots = _plan(lib, tmp_path, bdf.upper() if k == 1 else bdf)      # (hip prints upper-case hex on some stacks)
            assert (nd, slot, nslots) == (node, k, 4)
            assert len(cpus) == 32 and not (cpus & seen)              # an equal share, disjoint from every other rank's
            assert all((c + 128) in cpus for c in cpus if c < 128)    # a core's SMT sibling stays with it
            seen |= cpus
    assert seen == set(range(256))
    # unknown device / node -1: no binding, not an error
    d = tmp_path / "bus" / "pci" / "devices" / "0000:01:00.0"
    d.mkdir(parents=True)
    (d / "numa_node").write_text("-1\n")
    assert _plan(lib, tmp_path, "0000:01:00.0")[0] == set()
    assert _plan(lib, tmp_path, "0000:02:00.0")[0] == set()
