"""ctypes front-end of the CPU oracle (oracle/ba_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")



class _OptF64:
    """float64 array argument that may be None (NULL)."""

    @classmethod
    def from_param(cls, a):
        if a is None:
            return None
        return _f64p.from_param(a)


HIST_COLS = 8
STATUS_NAMES = {
    0: "Converged", 1: "MaxIterationsReached", 2: "CostToleranceReached",
    3: "ParameterToleranceReached", 4: "GradientToleranceReached", 5: "NumericalFailure",
    7: "Timeout", 8: "TrustRegionRadiusTooSmall", 9: "MinCostThresholdReached",
    11: "InvalidNumericalValues", 100: "LinearSolveFailed",
}


def build(native: bool = False) -> str:
    target = "libba_oracle_native.so" if native else "libba_oracle.so"
    subprocess.run(["make", "-C", _HERE, "native" if native else "all"], check=True,
                   stdout=subprocess.DEVNULL)
    return os.path.join(_HERE, target)


class LMConfig(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int), ("cost_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
        ("gradient_tolerance", C.c_double), ("damping", C.c_double), ("damping_min", C.c_double),
        ("damping_max", C.c_double), ("damping_nu", C.c_double), ("trust_region_radius", C.c_double),
        ("min_trust_region_radius", C.c_double), ("min_cost_threshold", C.c_double), ("variant", C.c_int),
        ("use_jacobi_scaling", C.c_int),
    ]

    @classmethod
    def default(cls, **kw) -> "LMConfig":
        """LevenbergMarquardtConfig::default (levenberg_marquardt.rs:318-358)."""
        c = cls(50, 1e-6, 1e-8, 1e-10, 1e-3, 1e-12, 1e12, 2.0, 1e4, 1e-32, -1.0, 0, 0)
        for k, v in kw.items():
            setattr(c, k, v)
        return c

    @classmethod
    def for_bundle_adjustment(cls, **kw) -> "LMConfig":
        """for_bundle_adjustment (:519-530): max 20 iterations, Iterative variant."""
        c = cls.default(max_iterations=20, variant=1)
        for k, v in kw.items():
            setattr(c, k, v)
        return c


_lib = None


def lib(native: bool = False):
    global _lib
    if _lib is not None and not native:
        return _lib
    path = os.path.join(_HERE, "libba_oracle_native.so" if native else "libba_oracle.so")
    src = os.path.join(_HERE, "ba_oracle.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        build(native)
    L = C.CDLL(path)
    vp = C.c_void_p
    L.ora_se3_plus.argtypes = [_f64p, _f64p, _f64p]
    L.ora_bal_project.argtypes = [_f64p, _f64p, _f64p]
    L.ora_bal_project.restype = C.c_int
    L.ora_bal_jacobian_point.argtypes = [_f64p, _f64p, _f64p]
    L.ora_bal_jacobian_intrinsics.argtypes = [_f64p, _f64p, _f64p]
    L.ora_huber_corrector.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.ora_huber_corrector.restype = C.c_double
    L.ora_linearize_obs.argtypes = [_f64p, _f64p, _f64p, _f64p, C.c_double, C.c_int, _f64p, _OptF64, _OptF64, _OptF64]
    L.ora_linearize_obs.restype = C.c_int
    L.ora_invert_landmark_blocks.argtypes = [C.c_int64, _f64p, C.c_double, _f64p]
    L.ora_invert_landmark_blocks.restype = C.c_int
    L.ora_schur_complement.argtypes = [C.c_int64, _f64p, C.c_int64, _i64p, _i64p, _f64p, _f64p, _f64p]
    L.ora_reduced_gradient.argtypes = [C.c_int64, _f64p, C.c_int64, _f64p, _i64p, _i64p, _f64p, _f64p, _f64p]
    L.ora_back_substitute.argtypes = [C.c_int64, _f64p, _f64p, _i64p, _i64p, _f64p, _f64p, _f64p]
    L.ora_solve_cholesky.argtypes = [C.c_int64, _f64p, _f64p, _f64p, C.c_void_p]
    L.ora_solve_cholesky.restype = C.c_int
    L.ora_solve_pcg.argtypes = [C.c_int64, _f64p, _f64p, C.c_int64, C.c_double, _f64p, C.POINTER(C.c_int64)]
    L.ora_solve_pcg.restype = C.c_int
    L.ora_schur_solve_dense_jacobian.argtypes = [C.c_int64, C.c_int64, C.c_int64, _f64p, _f64p, C.c_double,
                                                 C.c_int, C.c_int, C.c_double, _f64p, _f64p]
    L.ora_schur_solve_dense_jacobian.restype = C.c_int
    L.ora_update_damping.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_double]
    L.ora_update_damping.restype = C.c_int
    L.ora_step_quality.argtypes = [C.c_double, C.c_double, C.c_double]
    L.ora_step_quality.restype = C.c_double
    L.ora_compute_cost.argtypes = [C.c_int64, _f64p]
    L.ora_compute_cost.restype = C.c_double
    L.ora_problem_create.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int, _u32p, _u32p, _f64p,
                                     _i64p, _i64p, _i64p, C.c_double, vp, vp, vp]
    L.ora_problem_create.restype = vp
    L.ora_problem_destroy.argtypes = [vp]
    L.ora_set_params.argtypes = [vp, _f64p, _f64p, _f64p]
    L.ora_get_params.argtypes = [vp, _f64p, _f64p, _f64p]
    L.ora_set_cg_params.argtypes = [vp, C.c_int, C.c_double]
    L.ora_residuals.argtypes = [vp, vp]
    L.ora_residuals.restype = C.c_double
    L.ora_linearize.argtypes = [vp, vp, vp, vp, vp]
    L.ora_linearize.restype = C.c_double
    L.ora_solve_augmented.argtypes = [vp, C.c_double, C.c_int, _f64p, _f64p, vp, vp]
    L.ora_solve_augmented.restype = C.c_int
    L.ora_solve_augmented_quad.argtypes = [vp, C.c_double, _f64p, _OptF64]
    L.ora_solve_augmented_quad.restype = C.c_int
    L.ora_set_linearization.argtypes = [vp, _f64p, _f64p, _f64p, _OptF64]
    L.ora_column_norms.argtypes = [vp, _f64p]
    L.ora_column_norms.restype = C.c_int
    L.ora_set_column_scaling.argtypes = [vp, _OptF64]
    L.ora_set_column_scaling.restype = C.c_int
    L.ora_last_pcg_iters.argtypes = [vp]
    L.ora_last_pcg_iters.restype = C.c_int64
    L.ora_last_reg.argtypes = [vp]
    L.ora_last_reg.restype = C.c_double
    L.ora_last_sparse_stats.argtypes = [vp, _f64p]
    L.ora_last_sparse_stats.restype = None
    L.ora_solve_cholesky_sparse.argtypes = [C.c_int64, C.c_int64, _f64p, _f64p, _f64p, vp, vp]
    L.ora_solve_cholesky_sparse.restype = C.c_int
    L.ora_apply_step.argtypes = [vp, _f64p, C.c_double]
    L.ora_apply_step.restype = C.c_double
    L.ora_parameter_norm.argtypes = [vp]
    L.ora_parameter_norm.restype = C.c_double
    L.ora_lm_optimize.argtypes = [vp, C.POINTER(LMConfig), vp, C.c_int, C.POINTER(C.c_int),
                                  C.POINTER(C.c_double), C.POINTER(C.c_double), vp]
    L.ora_lm_optimize.restype = C.c_int
    if not native:
        _lib = L
    return L


MODES = {"ba": 0, "selfcal": 1, "only_pose": 2, "only_landmarks": 3, "only_intrinsics": 4, "pose_and_intrinsics": 5,
         "landmarks_and_intrinsics": 6}


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


@dataclass
class LMResult:
    status: str
    iterations: int
    initial_cost: float
    final_cost: float
    history: np.ndarray  # (iterations, 8): cost, damping, rho, accepted, |g|, |step|, predicted, trial cost
    steps: np.ndarray | None = None


@dataclass
class OracleProblem:
    """The reference's BA problem (bin/bundle_adjustment.rs) evaluated by the C oracle.

    mode: "ba" = OptimizationType::BundleAdjustment (keys [pose,pt]);
          "selfcal" = SelfCalibration (keys [pose,pt,intr], the reference's default).
    """

    n_cam: int
    n_pt: int
    cam_idx: np.ndarray
    pt_idx: np.ndarray
    obs_uv: np.ndarray
    intr_col: np.ndarray
    pose_col: np.ndarray
    pt_col: np.ndarray
    mode: str = "selfcal"
    huber_delta: float = 1.0
    fix_pose: np.ndarray | None = None
    fix_intr: np.ndarray | None = None
    fix_pt: np.ndarray | None = None
    native: bool = False
    _h: int = field(default=0, repr=False)

    def __post_init__(self):
        self._L = lib(self.native)
        self.cam_idx = np.ascontiguousarray(self.cam_idx, dtype=np.uint32)
        self.pt_idx = np.ascontiguousarray(self.pt_idx, dtype=np.uint32)
        self.obs_uv = np.ascontiguousarray(self.obs_uv, dtype=np.float64)
        self.intr_col = np.ascontiguousarray(self.intr_col, dtype=np.int64)
        self.pose_col = np.ascontiguousarray(self.pose_col, dtype=np.int64)
        self.pt_col = np.ascontiguousarray(self.pt_col, dtype=np.int64)
        self.n_obs = int(self.cam_idx.shape[0])
        self.cam_dof = 9 * self.n_cam
        self.total_dof = 9 * self.n_cam + 3 * self.n_pt
        fp = None if self.fix_pose is None else np.ascontiguousarray(self.fix_pose, dtype=np.uint8)
        fi = None if self.fix_intr is None else np.ascontiguousarray(self.fix_intr, dtype=np.uint8)
        ft = None if self.fix_pt is None else np.ascontiguousarray(self.fix_pt, dtype=np.uint8)
        self._keep = (fp, fi, ft)
        self._h = self._L.ora_problem_create(
            self.n_cam, self.n_pt, self.n_obs, MODES[self.mode],
            self.cam_idx, self.pt_idx, self.obs_uv, self.intr_col, self.pose_col, self.pt_col,
            float(self.huber_delta), _ptr(fp), _ptr(fi), _ptr(ft))

    def __del__(self):
        try:
            if self._h:
                self._L.ora_problem_destroy(self._h)
                self._h = 0
        except Exception:
            pass

    # -- parameters ---------------------------------------------------------
    def set_params(self, poses, intr, points):
        self._L.ora_set_params(self._h, np.ascontiguousarray(poses, dtype=np.float64),
                               np.ascontiguousarray(intr, dtype=np.float64),
                               np.ascontiguousarray(points, dtype=np.float64))

    def get_params(self):
        poses = np.empty((self.n_cam, 7)); intr = np.empty((self.n_cam, 3)); pts = np.empty((self.n_pt, 3))
        self._L.ora_get_params(self._h, poses, intr, pts)
        return poses, intr, pts

    def set_cg_params(self, max_iter: int, tol: float):
        self._L.ora_set_cg_params(self._h, int(max_iter), float(tol))

    # -- hot path -------------------------------------------------------------
    def residuals(self):
        r = np.empty(2 * self.n_obs)
        cost = self._L.ora_residuals(self._h, _ptr(r))
        return cost, r

    def linearize(self):
        r = np.empty(2 * self.n_obs)
        Jp = np.empty((self.n_obs, 2, 6)); Jl = np.empty((self.n_obs, 2, 3)); Ji = np.empty((self.n_obs, 2, 3))
        cost = self._L.ora_linearize(self._h, _ptr(r), _ptr(Jp), _ptr(Jl), _ptr(Ji))
        return cost, r, Jp, Jl, Ji

    def solve_augmented(self, lam: float, variant: int = 0, want_schur: bool = False):
        step = np.zeros(self.total_dof); grad = np.zeros(self.total_dof)
        S = np.empty((self.cam_dof, self.cam_dof)) if want_schur else None
        gred = np.empty(self.cam_dof) if want_schur else None
        rc = self._L.ora_solve_augmented(self._h, float(lam), int(variant), step, grad, _ptr(S), _ptr(gred))
        if rc != 0:
            raise RuntimeError(f"oracle solve_augmented failed: {rc}")
        return (step, grad, S, gred) if want_schur else (step, grad)

    def solve_augmented_quad(self, lam: float):
        """REFEREE (ba_oracle.c, ora_solve_augmented_quad): the exact step of the last linearisation's damped normal
        equations -- H, Hll^-1, S, g_red accumulated in __float128, the solve refined with __float128 residuals -- rounded
        to fp64 once.  Returns (step, info) with info = dict(sweeps, residual, fp64_forward_error, regularised_blocks)."""
        step = np.zeros(self.total_dof); info = np.zeros(4)
        rc = self._L.ora_solve_augmented_quad(self._h, float(lam), step, info)
        if rc != 0:
            raise RuntimeError(f"oracle referee failed: {rc}")
        return step, dict(sweeps=int(info[0]), residual=float(info[1]), fp64_forward_error=float(info[2]),
                          regularised_blocks=int(info[3]))

    def set_linearization(self, r, Jpose, Jpt, Jintr=None):
        """Replace the stored linearisation (e.g. by the blocks a device exported): the referee then judges a solver on
        its own equations.  Shapes: r (n_obs, 2), Jpose (n_obs, 2, 6), Jpt / Jintr (n_obs, 2, 3)."""
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        self._L.ora_set_linearization(self._h, f(r), f(Jpose), f(Jpt), None if Jintr is None else f(Jintr))

    def column_norms(self) -> np.ndarray:
        """compute_column_norms of the last linearisation (linearizer/mod.rs:229-239)."""
        n = np.empty(self.total_dof)
        self._L.ora_column_norms(self._h, n)
        return n

    def set_column_scaling(self, scaling):
        """J -> J diag(scaling) for the following solves (None: off); solve_augmented then returns the
        scaled step and gradient, as the reference's solver does for a scaled Jacobian."""
        self._L.ora_set_column_scaling(self._h, None if scaling is None else np.ascontiguousarray(scaling, dtype=np.float64))

    def apply_step(self, step, sign: float = 1.0) -> float:
        return self._L.ora_apply_step(self._h, np.ascontiguousarray(step, dtype=np.float64), float(sign))

    def parameter_norm(self) -> float:
        return self._L.ora_parameter_norm(self._h)

    @property
    def last_pcg_iters(self) -> int:
        return int(self._L.ora_last_pcg_iters(self._h))

    @property
    def last_reg(self) -> float:
        return float(self._L.ora_last_reg(self._h))

    def last_sparse_stats(self) -> dict:
        """Of the last variant-3 solve (ora_solve_cholesky_sparse): the sparsified S and its envelope after the ordering."""
        o = np.zeros(3); self._L.ora_last_sparse_stats(self._h, o)
        return dict(nonzero_blocks=int(o[0]), envelope_entries=float(o[1]), half_bandwidth=int(o[2]))

    def optimize(self, cfg: LMConfig, keep_steps: bool = False) -> LMResult:
        rows = cfg.max_iterations + 2
        hist = np.zeros((rows, HIST_COLS))
        steps = np.zeros((rows, self.total_dof)) if keep_steps else None
        it = C.c_int(0); c0 = C.c_double(0); c1 = C.c_double(0)
        st = self._L.ora_lm_optimize(self._h, C.byref(cfg), _ptr(hist), rows, C.byref(it),
                                     C.byref(c0), C.byref(c1), _ptr(steps))
        n = it.value
        return LMResult(STATUS_NAMES.get(st, str(st)), n, c0.value, c1.value, hist[:n].copy(),
                        None if steps is None else steps[:n].copy())


def from_data(data, layout, mode="selfcal", huber_delta=1.0, fix_first_pose=True, native=False) -> OracleProblem:
    """Build the oracle problem the way bin/bundle_adjustment.rs builds the reference's:
    all six DOF of pose_0000 fixed (:296-298), Huber(1.0) on every factor (:425-428)."""
    fix_pose = np.zeros((data.n_cam, 6), dtype=np.uint8)
    if fix_first_pose:
        fix_pose[0, :] = 1
    p = OracleProblem(data.n_cam, data.n_pt, data.cam_idx, data.pt_idx, data.obs_uv,
                      layout.intr_col, layout.pose_col, layout.pt_col, mode=mode,
                      huber_delta=huber_delta, fix_pose=fix_pose, native=native)
    p.set_params(data.poses, data.intr, data.points)
    return p
