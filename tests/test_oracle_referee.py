"""The __float128 referee of the oracle (ba_oracle.c, ora_solve_augmented_quad) -- CPU only.

The referee supplies the EXACT step of a linearisation's damped normal equations; the GPU tests use it to show that the
device step is as close to that exact step as the fp64 CPU path is (cond(S) is 1e9..1e10 on gauge-free bundle adjustment,
so two correct fp64 solvers differ by far more than the north star's 1e-10).  Here the referee itself is pinned:
  * against an independent 50-digit solve (mpmath) of the FULL damped normal equations built from the oracle's Jacobian
    blocks -- no Schur complement, no Cholesky, no shared code;
  * against the committed fixtures (it*_step_exact, it*_wc_*), which also hold the well-conditioned solves.
"""
import glob
import os

import numpy as np
import pytest

import apex_solver_amd as pkg
import np_ref
from apex_solver_amd.solver import OptimizationType, Problem
from test_oracle_golden import problem_from_golden, rel

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ba*.npz")))


def mp_direct_step(J, r, lam, digits=50):
    """(J^T J + lam I) x = -J^T r in `digits`-digit arithmetic (dense LU with pivoting, mpmath)."""
    import mpmath as mp

    mp.mp.dps = digits
    J = np.asarray(J.todense()) if hasattr(J, "todense") else np.asarray(J)
    m, n = J.shape
    Jm = mp.matrix(J.tolist())
    H = Jm.T * Jm
    for i in range(n):
        H[i, i] += mp.mpf(lam)
    g = Jm.T * mp.matrix([float(v) for v in r])
    x = mp.lu_solve(H, -g)
    return np.array([float(v) for v in x])


@pytest.mark.parametrize("mode", ["selfcal", "ba"])
def test_referee_vs_50_digit_direct_solve(oracle, mode):
    d = pkg.synthetic.make_problem(5, 24, 3, 5, config_id=77)
    ot = OptimizationType.SelfCalibration if mode == "selfcal" else OptimizationType.BundleAdjustment
    prob = Problem.bundle_adjustment(d, ot, 1.0)
    o = oracle.from_data(d, prob.layout, mode=mode)
    c, r, Jp, Jl, Ji = o.linearize()
    J = np_ref.sparse_jacobian(Jp, Jl, Ji, d.cam_idx, d.pt_idx, prob.layout, selfcal=(mode == "selfcal"))
    for lam in (1e-3, 1e4):
        exact, info = o.solve_augmented_quad(lam)
        x = mp_direct_step(J, r.ravel(), lam)
        step64 = o.solve_augmented(lam, 0)[0]
        print(mode, lam, "referee vs mpmath", rel(exact, x), "fp64 oracle vs mpmath", rel(step64, x), info)
        assert info["residual"] < 1e-28
        assert rel(exact, x) < 5e-16            # both are the exact step rounded to fp64 once
        assert rel(step64, x) < (1e-7 if lam < 1 else 1e-13)


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_referee_matches_golden(oracle, path):
    g = np.load(path)
    p = problem_from_golden(oracle, g)
    for it in range(int(g["iters"])):
        p.set_params(g[f"it{it}_poses"], g[f"it{it}_intr"], g[f"it{it}_points"])
        p.linearize()
        exact, info = p.solve_augmented_quad(float(g[f"it{it}_lambda"]))
        assert rel(exact, g[f"it{it}_step_exact"]) < 1e-15 and info["residual"] < 1e-28
        # the fp64 path against the exact step: this is the yardstick the device is held to (tests/test_gpu_parity.py)
        e64 = rel(g[f"it{it}_step"], exact)
        assert e64 < 1e-7
        # well-conditioned regime (lambda = 1e4, cond(S) <= 1e5): fp64 and exact agree far inside 1e-10
        wl = float(g[f"it{it}_wc_lambda"])
        wstep, _, wS, wgred = p.solve_augmented(wl, 0, want_schur=True)
        wexact, winfo = p.solve_augmented_quad(wl)
        assert np.linalg.cond(wS) < 1e5
        assert rel(wstep, g[f"it{it}_wc_step"]) < 1e-14 and rel(wexact, g[f"it{it}_wc_step_exact"]) < 1e-15
        assert rel(wstep, wexact) < 1e-13


def test_referee_on_a_foreign_linearisation(oracle):
    """ora_set_linearization: the referee judges a solver on the equations that solver built.  Perturbing J by 1e-13
    moves the exact step by far more than 1e-13 on an ill-conditioned system -- which is why a device is refereed on its
    own exported blocks as well as on the oracle's."""
    d = pkg.synthetic.make_problem(8, 60, 3, 6, config_id=78)
    prob = Problem.bundle_adjustment(d, OptimizationType.SelfCalibration, 1.0)
    o = oracle.from_data(d, prob.layout, mode="selfcal")
    c, r, Jp, Jl, Ji = o.linearize()
    e0, _ = o.solve_augmented_quad(1e-3)
    rng = np.random.default_rng(5)
    o.set_linearization(r, Jp * (1 + 1e-13 * rng.standard_normal(Jp.shape)), Jl, Ji)
    e1, _ = o.solve_augmented_quad(1e-3)
    o.set_linearization(r, Jp, Jl, Ji)
    e2, _ = o.solve_augmented_quad(1e-3)
    assert np.array_equal(e0, e2)
    assert 1e-13 < rel(e1, e0) < 1e-6
