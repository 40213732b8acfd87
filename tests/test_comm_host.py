"""The multi-rank communicator's shared-memory transport (apex-solver_amd/csrc/comm.cpp) WITHOUT a GPU: a host build
(tests/comm_host_harness.cpp, g++, -DAPEX_COMM_HOST_ONLY) driven by 2-4 real processes.  On the GPU box the same code runs
under the solver (tests/test_gpu_shm_ranks.py); here the transport itself is pinned: rendezvous, barrier, multi-round
transfers, rank-ordered (bitwise reproducible) sums, reduce-to-root, broadcast, all-gather -- and a segment left behind by
a crashed run under the same name (round-3 advice: the attach rendezvous passed at once on stale counters)."""
import ctypes as C
import os
import struct
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libcomm_host.so")
    src = os.path.join(ROOT, "tests", "comm_host_harness.cpp")
    deps = [src, os.path.join(ROOT, "apex-solver_amd", "csrc", "comm.cpp"), os.path.join(ROOT, "apex-solver_amd", "csrc", "comm.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread", src, "-o", so, "-lrt"], check=True)
    return so


WORKER = """
import ctypes as C, sys
L = C.CDLL(sys.argv[1])
L.comm_host_selftest.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_long, C.c_char_p, C.c_int]
msg = C.create_string_buffer(512)
rc = L.comm_host_selftest(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4].encode(), int(sys.argv[5]), msg, 512)
print(msg.value.decode())
sys.exit(rc)
"""


def _run(lib_path, world, name, n_big, stagger=()):
    import time

    procs = []
    order = list(stagger) + [r for r in range(world) if r not in stagger]
    for r in order:
        procs.append((r, subprocess.Popen([sys.executable, "-c", WORKER, lib_path, str(world), str(r), name, str(n_big)],
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        if r in stagger:
            time.sleep(0.3)   # the listed ranks get a head start (they must wait for rank 0's segment, not create one)
    outs = []
    for r, p in procs:
        o, _ = p.communicate(timeout=180)
        outs.append((r, p.returncode, o.strip()))
    assert all(rc == 0 for _, rc, _ in outs), outs
    assert not os.path.exists("/dev/shm/apexgpu-" + name), "rank 0 unlinks the name once everyone is attached"
    return outs


@pytest.mark.parametrize("world", [2, 3, 4])
def test_shm_collectives_between_processes(lib_path, world):
    # 1.3 M doubles = 10.4 MB per rank: three rounds through the 4 MiB slots
    outs = _run(lib_path, world, f"pytest-{os.getpid()}-w{world}", 1_300_003)
    assert all("ok (host shared memory)" in o for _, _, o in outs)


def test_shm_single_rank_and_late_rank_zero(lib_path):
    _run(lib_path, 1, f"pytest-{os.getpid()}-solo", 1000)
    # ranks 1 and 2 start before rank 0: they may not create the segment, only wait for it
    _run(lib_path, 3, f"pytest-{os.getpid()}-late0", 70_000, stagger=(1, 2))


def test_shm_stale_segment_of_a_crashed_run(lib_path):
    """A segment of the right size left under the same name, header saying 'ready, everyone attached, one rank waiting in
    a barrier': rank 0 replaces it, and a rank that mapped the leftover first moves over to the fresh one."""
    world = 2
    name = f"pytest-{os.getpid()}-stale"
    path = "/dev/shm/apexgpu-" + name
    size = 64 + world * (4 << 20)
    with open(path, "wb") as f:
        f.write(struct.pack("<5I", 1, 7, world, 0x41504558, 0) + b"\0" * 44)   # arrive, gen, attached, ready, go
        f.truncate(size)
    try:
        _run(lib_path, world, name, 50_000, stagger=(1,))
    finally:
        if os.path.exists(path):
            os.unlink(path)
