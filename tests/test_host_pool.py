"""apex-solver_amd/csrc/host_parallel.h on the CPU: the C ABI promises that no C++ exception crosses it (capi.cpp,
guarded()); set_structure runs its list construction on HostPool, so a throw inside a pooled loop body must come back on
the CALLING thread with the pool intact (round-2 advisor finding: it used to reach std::terminate in a worker)."""
import ctypes as C
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hp():
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libhost_pool_harness.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread", "-I", os.path.join(ROOT, "apex-solver_amd", "csrc"),
                    os.path.join(ROOT, "tests", "host_pool_harness.cpp"), "-o", so], check=True)
    L = C.CDLL(so)
    L.hp_throw.argtypes = [C.c_long, C.c_long, C.c_int, C.POINTER(C.c_long), C.POINTER(C.c_int)]
    L.hp_nested.argtypes = [C.c_long, C.c_long]
    L.hp_nested.restype = C.c_long
    L.hp_threads.restype = C.c_uint
    return L


@pytest.mark.parametrize("which,expect", [(0, 1), (1, 2)])
@pytest.mark.parametrize("bad", [0, 777, 3999])
def test_exception_in_a_pooled_loop_reaches_the_caller(hp, which, expect, bad):
    if hp.hp_threads() < 2:
        pytest.skip("single hardware thread: loops run serially")
    done = C.c_long(0); ok = C.c_int(0)
    for _ in range(20):     # (which participant hits the bad row varies from run to run: worker or caller)
        assert hp.hp_throw(4000, bad, which, C.byref(done), C.byref(ok)) == expect
        assert done.value < 4000 and ok.value == 1


def test_no_exception_visits_every_row(hp):
    done = C.c_long(0); ok = C.c_int(0)
    assert hp.hp_throw(4000, -1, 0, C.byref(done), C.byref(ok)) == 0
    assert done.value == 4000 and ok.value == 1


def test_nested_loops_of_other_instantiations_run_serially(hp):
    # 64 outer rows x (100 + 50): would self-deadlock on the pool's non-recursive mutex with a per-template flag
    assert hp.hp_nested(64, 100) == 64 * 150
