"""The per-edge device math of the pose-graph kernels (apex-solver_amd/csrc/pg_device.hpp), compiled for
the host, against the oracle.  No GPU needed: isolates formula errors from kernel-structure errors."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import apex_solver_amd as pkg
from oracle import pg_oracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_f = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


@pytest.fixture(scope="module")
def hh():
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libhost_harness_pg.so")
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC",
                    "-I", os.path.join(ROOT, "apex-solver_amd", "csrc"),
                    os.path.join(ROOT, "tests", "host_harness.cpp"), "-o", so], check=True)
    L = C.CDLL(so)
    L.hh_between_linearize.argtypes = [_f, _f, _f, C.c_double, _f, _f]
    L.hh_between_normal.argtypes = [_f] * 8
    return L


def cases():
    d = pkg.synthetic.make_sphere(12, 12)
    rng = np.random.default_rng(0)
    out = [(d.poses[d.e_from[e]], d.poses[d.e_to[e]], d.meas[e]) for e in range(0, d.n_e, 7)]
    ident = np.array([0, 0, 0, 1.0, 0, 0, 0])
    out.append((ident, ident, ident))                      # zero residual: the small-angle branches
    for _ in range(20):                                    # large rotations, incl. w < 0 quaternions
        q = rng.normal(size=(3, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
        t = rng.uniform(-5, 5, size=(3, 3))
        out.append(tuple(np.concatenate([t[i], q[i]]) for i in range(3)))
    k = np.array([1.0, 2, 3, 2.0, 0.2, -0.4, 0.6])         # un-normalised input quaternion
    out.append((k, ident, ident))
    return out


def test_between_linearize_matches_oracle(hh):
    worst_r = worst_j = 0.0
    for k0, k1, m in cases():
        for delta in (-1.0, 0.5):
            r = np.zeros(6); J = np.zeros((6, 12))
            hh.hh_between_linearize(np.ascontiguousarray(k0), np.ascontiguousarray(k1), np.ascontiguousarray(m), delta, r, J)
            ro, Jo = po.between_linearize(k0, k1, m)
            if delta > 0:
                s = float(ro @ ro)
                sc = np.sqrt(delta / np.sqrt(s)) if s > delta * delta else 1.0
                ro, Jo = ro * sc, Jo * sc
            worst_r = max(worst_r, np.abs(r - ro).max() / max(1.0, np.abs(ro).max()))
            worst_j = max(worst_j, np.abs(J - Jo).max() / max(1.0, np.abs(Jo).max()))
    assert worst_r < 1e-13 and worst_j < 1e-12, (worst_r, worst_j)


def test_structured_normal_products_match_dense(hh):
    for k0, k1, m in cases()[:40]:
        H00 = np.zeros((6, 6)); H11 = np.zeros((6, 6)); H10 = np.zeros((6, 6)); g0 = np.zeros(6); g1 = np.zeros(6)
        hh.hh_between_normal(np.ascontiguousarray(k0), np.ascontiguousarray(k1), np.ascontiguousarray(m), H00, H11, H10, g0, g1)
        r, J = po.between_linearize(k0, k1, m)
        J0, J1 = J[:, :6], J[:, 6:]
        sc = max(1.0, np.abs(J).max() ** 2)
        assert np.abs(H00 - J0.T @ J0).max() < 1e-12 * sc
        assert np.abs(H11 - J1.T @ J1).max() < 1e-12 * sc
        assert np.abs(H10 - J1.T @ J0).max() < 1e-12 * sc
        assert np.abs(g0 - J0.T @ r).max() < 1e-12 * sc and np.abs(g1 - J1.T @ r).max() < 1e-12 * sc
