"""ctypes front-end of the pose-graph CPU oracle (oracle/pg_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_f64 = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i64 = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_u8 = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


class _Opt:
    @classmethod
    def from_param(cls, a):
        return None if a is None else _f64.from_param(a)


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.join(_HERE, "libpg_oracle.so")
    if not os.path.exists(path):
        subprocess.run(["make", "-C", _HERE, "libpg_oracle.so"], check=True, stdout=subprocess.DEVNULL)
    L = C.CDLL(path)
    for name, n_in in (("pgo_so3_log", 1), ("pgo_so3_exp", 1), ("pgo_so3_left_jacobian", 1), ("pgo_so3_left_jacobian_inv", 1),
                       ("pgo_se3_log", 1), ("pgo_se3_exp", 1), ("pgo_se3_adjoint", 1), ("pgo_se3_inverse", 1),
                       ("pgo_se3_right_jacobian", 1), ("pgo_se3_left_jacobian", 1), ("pgo_se3_right_jacobian_inv", 1),
                       ("pgo_se3_left_jacobian_inv", 1), ("pgo_se3_q_block", 2), ("pgo_se3_compose", 2), ("pgo_se3_between", 2),
                       ("pgo_se3_plus", 2)):
        f = getattr(L, name)
        f.argtypes = [_f64] * (n_in + 1)
        f.restype = None
    L.pgo_between_linearize.argtypes = [_f64, _f64, _f64, _f64, _Opt]
    L.pgo_between_linearize.restype = None
    L.pgo_create.argtypes = [C.c_int64, C.c_int64, _i64, _i64, _f64, _i64, _u8, C.c_double]
    L.pgo_create.restype = C.c_void_p
    L.pgo_destroy.argtypes = [C.c_void_p]
    L.pgo_set_params.argtypes = [C.c_void_p, _f64]
    L.pgo_get_params.argtypes = [C.c_void_p, _f64]
    L.pgo_residuals.argtypes = [C.c_void_p, _Opt]
    L.pgo_residuals.restype = C.c_double
    L.pgo_linearize.argtypes = [C.c_void_p, _Opt, _Opt]
    L.pgo_linearize.restype = C.c_double
    L.pgo_normal_equations.argtypes = [C.c_void_p, _Opt, _f64]
    L.pgo_solve_augmented.argtypes = [C.c_void_p, C.c_double, _Opt, _Opt]
    L.pgo_solve_augmented.restype = C.c_int
    L.pgo_solve_dense_jacobian.argtypes = [C.c_int64, C.c_int64, _f64, _f64, C.c_double, _f64, _Opt]
    L.pgo_solve_dense_jacobian.restype = C.c_int
    L.pgo_column_norms.argtypes = [C.c_void_p, _f64]
    L.pgo_column_norms.restype = None
    L.pgo_set_column_scaling.argtypes = [C.c_void_p, _Opt]
    L.pgo_set_column_scaling.restype = None
    L.pgo_apply_step.argtypes = [C.c_void_p, _f64, C.c_double]
    L.pgo_parameter_norm.argtypes = [C.c_void_p]
    L.pgo_parameter_norm.restype = C.c_double
    L.pgo_lm_optimize.argtypes = [C.c_void_p, _f64, _Opt, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double),
                                  C.POINTER(C.c_double), _Opt]
    L.pgo_lm_optimize.restype = C.c_int
    L.pgo_set_priors.argtypes = [C.c_void_p, C.c_int64, _i64, _f64, _f64]
    L.pgo_set_priors.restype = None
    L.pgo_prior_residuals.argtypes = [C.c_void_p, _f64]
    L.pgo_prior_residuals.restype = None
    _lib = L
    return L


def call(name: str, *ins, out_shape):
    """Run one of the small group functions: inputs are float arrays, returns the output array."""
    L = lib()
    out = np.zeros(out_shape)
    getattr(L, name)(*[np.ascontiguousarray(a, dtype=np.float64) for a in ins], out)
    return out


def between_linearize(k0, k1, meas, want_jac=True):
    L = lib()
    r = np.zeros(6)
    J = np.zeros((6, 12)) if want_jac else None
    L.pgo_between_linearize(np.ascontiguousarray(k0, dtype=np.float64), np.ascontiguousarray(k1, dtype=np.float64),
                            np.ascontiguousarray(meas, dtype=np.float64), r, J)
    return r, J


def solve_dense_jacobian(J, r, lam=0.0):
    """(J^T J + lam I) dx = -J^T r with the oracle's LL^T; returns (rc, dx, grad)."""
    J = np.ascontiguousarray(J, dtype=np.float64); r = np.ascontiguousarray(r, dtype=np.float64)
    dx = np.zeros(J.shape[1]); g = np.zeros(J.shape[1])
    rc = lib().pgo_solve_dense_jacobian(J.shape[0], J.shape[1], J, r, float(lam), dx, g)
    return rc, dx, g


def lm_config(max_iterations=50, cost_tolerance=1e-6, parameter_tolerance=1e-8, gradient_tolerance=1e-10, damping=1e-3,
              damping_min=1e-12, damping_max=1e12, nu=2.0, trust_region_radius=1e4, min_trust_region_radius=1e-32,
              min_cost_threshold=-1.0, use_jacobi_scaling=False) -> np.ndarray:
    """LevenbergMarquardtConfig::default (levenberg_marquardt.rs:318-358) as the array pgo_lm_optimize takes."""
    return np.array([max_iterations, cost_tolerance, parameter_tolerance, gradient_tolerance, damping, damping_min,
                     damping_max, nu, trust_region_radius, min_trust_region_radius, min_cost_threshold,
                     1.0 if use_jacobi_scaling else 0.0], dtype=np.float64)


class PgOracle:
    """One pose-graph problem in the oracle (edges in residual-block order, caller's column offsets)."""

    def __init__(self, e_from, e_to, meas, pose_col, fix=None, huber_delta=None, poses=None):
        self.L = lib()
        self.n_v = int(len(pose_col)); self.n_e = int(len(e_from))
        fix = np.zeros((self.n_v, 6), np.uint8) if fix is None else np.ascontiguousarray(fix, dtype=np.uint8)
        self.p = self.L.pgo_create(self.n_v, self.n_e, np.ascontiguousarray(e_from, dtype=np.int64),
                                   np.ascontiguousarray(e_to, dtype=np.int64), np.ascontiguousarray(meas, dtype=np.float64),
                                   np.ascontiguousarray(pose_col, dtype=np.int64), fix,
                                   -1.0 if huber_delta is None else float(huber_delta))
        if poses is not None:
            self.set_params(poses)

    @classmethod
    def from_problem(cls, problem):
        d = problem.data
        o = cls(d.e_from, d.e_to, d.meas, problem.pose_col, problem.fix, problem.huber_delta, d.poses)
        if getattr(problem, "priors", None):
            o.set_priors([v for v, _, _ in problem.priors], [x for _, x, _ in problem.priors], [dl for _, _, dl in problem.priors])
        return o

    def set_priors(self, vertex, data7, huber_delta):
        """PriorFactor blocks (prior_factor.rs:96-108) on SE3 variables; huber_delta <= 0 / None: no loss."""
        v = np.ascontiguousarray(vertex, dtype=np.int64).reshape(-1)
        x = np.ascontiguousarray(data7, dtype=np.float64).reshape(-1, 7)
        dl = np.ascontiguousarray([-1.0 if q is None else float(q) for q in huber_delta], dtype=np.float64)
        assert len(v) == len(x) == len(dl)
        self.n_prior = len(v)
        self.L.pgo_set_priors(self.p, len(v), v, x, dl)

    def prior_residuals(self):
        """Corrected residuals of the prior blocks at the last residuals() / linearize(): [n_prior][7]."""
        r = np.zeros((getattr(self, "n_prior", 0), 7))
        if len(r): self.L.pgo_prior_residuals(self.p, r)
        return r

    def __del__(self):
        if getattr(self, "p", None):
            self.L.pgo_destroy(self.p)
            self.p = None

    def set_params(self, poses): self.L.pgo_set_params(self.p, np.ascontiguousarray(poses, dtype=np.float64))

    def get_params(self):
        out = np.zeros((self.n_v, 7))
        self.L.pgo_get_params(self.p, out)
        return out

    def residuals(self):
        r = np.zeros((self.n_e, 6))
        return self.L.pgo_residuals(self.p, r), r

    def linearize(self):
        r = np.zeros((self.n_e, 6)); J = np.zeros((self.n_e, 6, 12))
        c = self.L.pgo_linearize(self.p, r, J)
        return c, r, J

    def normal_equations(self, dense=True):
        n = 6 * self.n_v
        H = np.zeros((n, n)) if dense else None
        g = np.zeros(n)
        self.L.pgo_normal_equations(self.p, H, g)
        return H, g

    def solve_augmented(self, lam):
        n = 6 * self.n_v
        step = np.zeros(n); grad = np.zeros(n)
        rc = self.L.pgo_solve_augmented(self.p, float(lam), step, grad)
        return rc, step, grad

    def column_norms(self):
        n = np.zeros(6 * self.n_v)
        self.L.pgo_column_norms(self.p, n)
        return n

    def set_column_scaling(self, scaling):
        self.L.pgo_set_column_scaling(self.p, None if scaling is None else np.ascontiguousarray(scaling, dtype=np.float64))

    def apply_step(self, step, sign=1.0): self.L.pgo_apply_step(self.p, np.ascontiguousarray(step, dtype=np.float64), float(sign))
    def parameter_norm(self): return self.L.pgo_parameter_norm(self.p)

    def lm_optimize(self, cfg: np.ndarray, hist_rows=64, want_params=False):
        hist = np.zeros((hist_rows, 8))
        params = np.zeros((hist_rows, self.n_v, 7)) if want_params else None
        it = C.c_int(); c0 = C.c_double(); c1 = C.c_double()
        st = self.L.pgo_lm_optimize(self.p, cfg, hist, hist_rows, C.byref(it), C.byref(c0), C.byref(c1), params)
        n = min(it.value, hist_rows)
        return {"status": st, "iterations": it.value, "initial_cost": c0.value, "final_cost": c1.value,
                "history": hist[:n], "params": None if params is None else params[:n]}
