"""Host-side mirror of the reference's interface for the bundle-adjustment path.

Names, argument meaning and error behaviour follow apex-solver (Rust) so that the parity tests
read like the reference's own tests:

  Problem / fix_variable               src/core/problem.rs:465-489, 609-616 (BA subset: one BAL
                                       ProjectionFactor + HuberLoss per observation, as
                                       bin/bundle_adjustment.rs:232-298, 391-441 builds it)
  LinearSolverType, SchurVariant       src/linalg/mod.rs:48-57 ; explicit_schur.rs:58-65
  LevenbergMarquardtConfig             src/optimizer/levenberg_marquardt.rs:213-530
  LevenbergMarquardt.optimize          :1034-1083 (dispatch) and :823-1031 (loop; the loop itself
                                       runs in the library's C++ twin, apexgpu_lm_optimize)
  GpuSchurComplementSolver             LinearSolver<M> + StructureAware (src/linalg/mod.rs:116-180)
                                       as SparseSchurComplementSolver implements them
                                       (explicit_schur.rs:1038-1243)
  SolverResult, OptimizationStatus     src/optimizer/mod.rs:189-273

All numerics run in libapexgpu.so (HIP, gfx950); nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
import enum
from dataclasses import dataclass, field, replace

import numpy as np

from . import capi
from .layout import ColumnLayout, reference_column_layout
from .synthetic import BAProblemData


class LinearSolverType(enum.Enum):
    """src/linalg/mod.rs:48-57 (#[non_exhaustive]) + the variant this backend adds."""

    SparseCholesky = "SparseCholesky"
    SparseQR = "SparseQR"
    SparseSchurComplement = "SparseSchurComplement"
    DenseCholesky = "DenseCholesky"
    DenseQR = "DenseQR"
    GpuSchurComplement = "GpuSchurComplement"
    GpuSparseCholesky = "GpuSparseCholesky"


class SchurVariant(enum.Enum):
    Sparse = 0      # explicit S + Cholesky
    Iterative = 1   # explicit S + Jacobi-PCG (explicit_schur.rs:1117-1120)
    Implicit = 2    # this backend only: IterativeSchurSolver's matrix-free PCG (implicit_schur.rs), not reachable
                    # through the reference's LevenbergMarquardt (its Iterative arm forms S explicitly)


class SchurPreconditioner(enum.Enum):
    """Stored but ignored on the LM path, exactly like the reference."""

    None_ = 0
    BlockDiagonal = 1
    SchurJacobi = 2


class OptimizationType(enum.Enum):
    """OptimizeParams<POSE, LANDMARK, INTRINSIC> (src/factors/mod.rs:66-101; bin/bundle_adjustment.rs:268-292)."""

    BundleAdjustment = 0  # factor keys [pose, pt]
    SelfCalibration = 1   # factor keys [pose, pt, intr]  (the reference's default)
    OnlyPose = 2                # <true, false, false>: landmarks and intrinsics are constants of the factors
    OnlyLandmarks = 3           # <false, true, false>
    OnlyIntrinsics = 4          # <false, false, true>
    PoseAndIntrinsics = 5       # <true, false, true>
    LandmarksAndIntrinsics = 6  # <false, true, true>

    @property
    def flags(self):
        """(POSE, LANDMARK, INTRINSIC)"""
        return {0: (1, 1, 0), 1: (1, 1, 1), 2: (1, 0, 0), 3: (0, 1, 0), 4: (0, 0, 1), 5: (1, 0, 1), 6: (0, 1, 1)}[self.value]


class OptimizationStatus(enum.Enum):
    Converged = 0
    MaxIterationsReached = 1
    CostToleranceReached = 2
    ParameterToleranceReached = 3
    GradientToleranceReached = 4
    NumericalFailure = 5
    UserTerminated = 6
    Timeout = 7
    TrustRegionRadiusTooSmall = 8
    MinCostThresholdReached = 9
    IllConditionedJacobian = 10
    InvalidNumericalValues = 11
    LinearSolveFailed = 100  # OptimizerError::LinearSolveFailed aborts optimize() in the reference


@dataclass
class LevenbergMarquardtConfig:
    """Defaults of levenberg_marquardt.rs:318-358.  Fields the reference stores but never reads
    (min_relative_decrease, good_step_quality, ... SURVEY.md §3.2) are not reproduced."""

    linear_solver_type: LinearSolverType = LinearSolverType.GpuSchurComplement
    max_iterations: int = 50
    cost_tolerance: float = 1e-6
    parameter_tolerance: float = 1e-8
    gradient_tolerance: float = 1e-10
    timeout: float | None = None
    damping: float = 1e-3
    damping_min: float = 1e-12
    damping_max: float = 1e12
    damping_nu: float = 2.0
    trust_region_radius: float = 1e4
    min_trust_region_radius: float = 1e-32
    min_cost_threshold: float | None = None
    schur_variant: SchurVariant = SchurVariant.Sparse
    schur_preconditioner: SchurPreconditioner = SchurPreconditioner.None_
    use_jacobi_scaling: bool = False  # :352

    @classmethod
    def new(cls) -> "LevenbergMarquardtConfig":
        return cls()

    @classmethod
    def for_bundle_adjustment(cls) -> "LevenbergMarquardtConfig":
        """:519-530 -- Schur complement, Iterative variant, 20 iterations, Ceres tolerances."""
        return cls(max_iterations=20, schur_variant=SchurVariant.Iterative,
                   schur_preconditioner=SchurPreconditioner.SchurJacobi)

    # builder methods (:361-492)
    def with_linear_solver_type(self, t): return replace(self, linear_solver_type=t)
    def with_max_iterations(self, n): return replace(self, max_iterations=int(n))
    def with_cost_tolerance(self, v): return replace(self, cost_tolerance=float(v))
    def with_parameter_tolerance(self, v): return replace(self, parameter_tolerance=float(v))
    def with_gradient_tolerance(self, v): return replace(self, gradient_tolerance=float(v))
    def with_timeout(self, seconds): return replace(self, timeout=seconds)
    def with_damping(self, v): return replace(self, damping=float(v))
    def with_damping_bounds(self, lo, hi): return replace(self, damping_min=float(lo), damping_max=float(hi))
    def with_min_cost_threshold(self, v): return replace(self, min_cost_threshold=v)
    def with_schur_variant(self, v): return replace(self, schur_variant=v)
    def with_schur_preconditioner(self, v): return replace(self, schur_preconditioner=v)
    def with_jacobi_scaling(self, on): return replace(self, use_jacobi_scaling=bool(on))  # :474-477

    def to_c(self) -> capi.LmConfigC:
        return capi.LmConfigC(
            self.max_iterations, self.cost_tolerance, self.parameter_tolerance, self.gradient_tolerance,
            self.damping, self.damping_min, self.damping_max, self.damping_nu, self.trust_region_radius,
            self.min_trust_region_radius, -1.0 if self.min_cost_threshold is None else self.min_cost_threshold,
            -1.0 if self.timeout is None else float(self.timeout), self.schur_variant.value,
            1 if self.use_jacobi_scaling else 0)


@dataclass
class Problem:
    """The factor graph bin/bundle_adjustment.rs builds: variables `pose_{i:04}` (SE3),
    `intr_{i:04}` (Rn 3), `pt_{j:05}` (Rn 3); one ProjectionFactor<BALPinholeCameraStrict, OP>
    with HuberLoss(1.0) per observation.  Only what the device backend has to ingest is kept."""

    data: BAProblemData
    optimization_type: OptimizationType = OptimizationType.SelfCalibration
    huber_delta: float | None = 1.0
    fix_pose: np.ndarray = field(default=None)
    fix_intr: np.ndarray = field(default=None)
    fix_pt: np.ndarray = field(default=None)

    def __post_init__(self):
        d = self.data
        if self.fix_pose is None:
            self.fix_pose = np.zeros((d.n_cam, 6), dtype=np.uint8)
        if self.fix_intr is None:
            self.fix_intr = np.zeros((d.n_cam, 3), dtype=np.uint8)
        if self.fix_pt is None:
            self.fix_pt = np.zeros((d.n_pt, 3), dtype=np.uint8)
        self.layout: ColumnLayout = reference_column_layout(d.n_cam, d.n_pt)

    @classmethod
    def bundle_adjustment(cls, data: BAProblemData, optimization_type=OptimizationType.SelfCalibration,
                          huber_delta: float | None = 1.0) -> "Problem":
        """run_bundle_adjustment (bin/bundle_adjustment.rs:211-298): gauge fixed by all six DOF of
        pose_0000."""
        p = cls(data, optimization_type, huber_delta)
        for dof in range(6):
            p.fix_variable("pose_0000", dof)
        return p

    def fix_variable(self, name: str, dof: int):
        """Problem::fix_variable (src/core/problem.rs:609-616)."""
        kind, idx = name.split("_")
        i = int(idx)
        if kind == "pose":
            self.fix_pose[i, dof] = 1
        elif kind == "intr":
            self.fix_intr[i, dof] = 1
        elif kind == "pt":
            self.fix_pt[i, dof] = 1
        else:
            raise KeyError(name)

    @property
    def total_dof(self) -> int:
        return self.layout.total_dof

    @property
    def num_residual_blocks(self) -> int:
        return self.data.n_obs


class GpuSchurComplementSolver:
    """Device-resident explicit-Schur solver behind the reference's LinearSolver surface.

    Differences from SparseSchurComplementSolver that the boundary hides: the Jacobian is never
    materialised (the solver linearises the factors itself, so `solve_augmented_equation` takes
    only lambda), and the parameters live on the device between calls."""

    def __init__(self, device: int = 0):
        self.device = device
        self.variant = SchurVariant.Sparse
        self.cg_max_iterations = 200
        self.cg_tolerance = 1e-6
        self._h: capi.Handle | None = None
        self._problem: Problem | None = None
        self._gradient = None
        self._shard = None
        self._comm = None
        self._pre_options = {}

    # builder methods (explicit_schur.rs:219-238)
    def with_variant(self, v: SchurVariant): self.variant = v; return self
    def with_preconditioner(self, _p): return self  # ignored on this path, like the reference
    def with_cg_params(self, max_iter: int, tol: float):
        self.cg_max_iterations, self.cg_tolerance = int(max_iter), float(tol)
        if self._h is not None:
            self._h.check(self._h.L.apexgpu_set_cg_params(self._h.h, self.cg_max_iterations, self.cg_tolerance))
        return self
    def with_option(self, name: str, value: int):
        """An implementation switch that must be set before initialize_structure (e.g. nested_dissection)."""
        self._pre_options[name] = int(value); return self

    def with_shard(self, rank: int, world: int): self._shard = (rank, world); return self
    def with_communicator(self, world: int, rank: int, unique_id: bytes): self._comm = (world, rank, unique_id); return self
    def with_shm_communicator(self, world: int, rank: int, name: str):
        """The multi-rank schedule over host shared memory (bring-up / tests: csrc/comm.h) instead of RCCL."""
        self._comm = (world, rank, name); return self

    # StructureAware::initialize_structure
    def initialize_structure(self, problem: Problem):
        import time
        t0 = time.perf_counter()
        d = problem.data
        mode = problem.optimization_type.value
        self._h = capi.Handle(d.n_cam, d.n_pt, d.n_obs, mode, self.device)   # the process's first HIP call: runtime start-up
        h = self._h
        if self._comm is not None:
            world, rank, uid = self._comm
            if isinstance(uid, str):    # host shared-memory transport
                h.check(h.L.apexgpu_comm_init_shm(h.h, world, rank, uid.encode()))
            else:
                buf = (C.c_char * 128).from_buffer_copy(uid)
                h.check(h.L.apexgpu_comm_init(h.h, world, rank, C.cast(buf, C.c_void_p)))
        elif self._shard is not None:
            h.check(h.L.apexgpu_set_shard(h.h, *self._shard))
        for k, v in self._pre_options.items():
            h.check(h.L.apexgpu_set_option(h.h, k.encode(), v))
        t1 = time.perf_counter()
        lay = problem.layout
        self._keep = [np.ascontiguousarray(a) for a in (
            d.cam_idx.astype(np.uint32, copy=False), d.pt_idx.astype(np.uint32, copy=False), d.obs_uv.astype(np.float64, copy=False),
            lay.intr_col, lay.pose_col, lay.pt_col, problem.fix_pose, problem.fix_intr, problem.fix_pt)]
        hd = -1.0 if problem.huber_delta is None else float(problem.huber_delta)
        t2 = time.perf_counter()
        h.check(h.L.apexgpu_set_structure(h.h, *[capi.ptr(a) for a in self._keep], hd))
        t3 = time.perf_counter()
        h.check(h.L.apexgpu_set_cg_params(h.h, self.cg_max_iterations, self.cg_tolerance))
        self._problem = problem
        # wall time of this call by piece (seconds): handle + communicator, host-side argument arrays, apexgpu_set_structure
        self.setup_wall = dict(create_handle=t1 - t0, host_arrays=t2 - t1, set_structure=t3 - t2)
        return self

    def reinitialize_structure(self, problem: Problem):
        """A second initialize_structure on the SAME handle (StructureAware::initialize_structure may be called again,
        src/linalg/mod.rs:116-123): the counts are those the handle was created with, the lists and every decision that follows
        from them (tile plan, variant selection) are made afresh.  Options stay as they were set."""
        h = self._need()
        d, old = problem.data, self._problem.data
        if (d.n_cam, d.n_pt, d.n_obs) != (old.n_cam, old.n_pt, old.n_obs) or problem.optimization_type != self._problem.optimization_type:
            raise capi.LinAlgError(-5, "reinitialize_structure: the handle was created for other counts")
        lay = problem.layout
        self._keep = [np.ascontiguousarray(a) for a in (
            d.cam_idx.astype(np.uint32, copy=False), d.pt_idx.astype(np.uint32, copy=False), d.obs_uv.astype(np.float64, copy=False),
            lay.intr_col, lay.pose_col, lay.pt_col, problem.fix_pose, problem.fix_intr, problem.fix_pt)]
        hd = -1.0 if problem.huber_delta is None else float(problem.huber_delta)
        h.check(h.L.apexgpu_set_structure(h.h, *[capi.ptr(a) for a in self._keep], hd))
        h.check(h.L.apexgpu_set_cg_params(h.h, self.cg_max_iterations, self.cg_tolerance))
        self._problem = problem
        return self

    def _need(self) -> capi.Handle:
        if self._h is None:
            raise capi.LinAlgError(-5, "Block structure not built. Call initialize_structure() first.")
        return self._h

    def set_parameters(self, poses, intr, points):
        import time
        t0 = time.perf_counter()
        h = self._need()
        a = [np.ascontiguousarray(x, dtype=np.float64) for x in (poses, intr, points)]
        h.check(h.L.apexgpu_set_params(h.h, *[capi.ptr(x) for x in a]))
        if hasattr(self, "setup_wall"): self.setup_wall["set_parameters"] = time.perf_counter() - t0

    def get_parameters(self):
        h = self._need()
        poses = np.empty((h.n_cam, 7)); intr = np.empty((h.n_cam, 3)); pts = np.empty((h.n_pt, 3))
        h.check(h.L.apexgpu_get_params(h.h, capi.ptr(poses), capi.ptr(intr), capi.ptr(pts)))
        return poses, intr, pts

    def compute_cost(self) -> float:
        h = self._need(); c = C.c_double()
        h.check(h.L.apexgpu_cost(h.h, C.byref(c)))
        return c.value

    def assemble(self, lam: float):
        """A1-A11 only: S, g_red, H_ll^-1, g on the device at the current parameters."""
        h = self._need()
        h.check(h.L.apexgpu_assemble(h.h, float(lam)))

    # LinearSolver::solve_augmented_equation / solve_normal_equation
    def solve_augmented_equation(self, lam: float, want_step: bool = True):
        h = self._need()
        n = self._problem.total_dof
        step = np.zeros(n) if want_step else None
        grad = np.zeros(n) if want_step else None
        h.check(h.L.apexgpu_solve_augmented(h.h, float(lam), self.variant.value, capi.ptr(step), capi.ptr(grad)))
        self._gradient = grad
        return step

    def solve_normal_equation(self):
        return self.solve_augmented_equation(0.0)

    def get_gradient(self):
        """+J^T r of the last solve (explicit_schur.rs:1240-1242); None before any solve."""
        return self._gradient

    # AssemblyBackend::compute_column_norms / apply_column_scaling / apply_inverse_scaling (linearizer/mod.rs:229-262)
    def compute_column_norms(self) -> np.ndarray:
        h = self._need()
        n = np.zeros(self._problem.total_dof)
        h.check(h.L.apexgpu_column_norms(h.h, capi.ptr(n)))
        return n

    def apply_column_scaling(self, scaling):
        """J -> J diag(scaling) for the following solves (None: off).  solve_augmented_equation then returns the
        scaled step and get_gradient the scaled gradient, like the reference's solver fed with the scaled Jacobian."""
        h = self._need()
        a = None if scaling is None else np.ascontiguousarray(scaling, dtype=np.float64)
        if a is not None and a.shape != (self._problem.total_dof,):
            raise ValueError("scaling must have total_dof entries")
        h.check(h.L.apexgpu_set_column_scaling(h.h, capi.ptr(a)))
        self._scaling = a

    def apply_inverse_scaling(self, step):
        s = getattr(self, "_scaling", None)
        return step if s is None else step * s

    def step_stats(self):
        h = self._need(); out = (C.c_double * 3)()
        h.check(h.L.apexgpu_step_stats(h.h, C.byref(out)))
        return tuple(out)

    def eval_step(self) -> float:
        h = self._need(); c = C.c_double()
        h.check(h.L.apexgpu_eval_step(h.h, C.byref(c)))
        return c.value

    def commit_step(self): h = self._need(); h.check(h.L.apexgpu_commit_step(h.h))
    def discard_step(self): h = self._need(); h.check(h.L.apexgpu_discard_step(h.h))

    def parameter_norm(self) -> float:
        h = self._need(); c = C.c_double()
        h.check(h.L.apexgpu_parameter_norm(h.h, C.byref(c)))
        return c.value

    # parity / debug
    def get_residual(self):
        h = self._need(); r = np.zeros(2 * h.n_obs)
        h.check(h.L.apexgpu_get_residual(h.h, capi.ptr(r)))
        return r

    def get_jacobian_blocks(self):
        h = self._need(); dc = 9 if h.mode in (1, 4, 5, 6) else 6
        jc = np.zeros((h.n_obs, 2, dc)); jl = np.zeros((h.n_obs, 2, 3))
        h.check(h.L.apexgpu_get_jacobian_blocks(h.h, capi.ptr(jc), capi.ptr(jl)))
        return jc, jl

    def get_schur(self, want_S: bool = True):
        h = self._need(); n = 9 * h.n_cam
        S = np.zeros((n, n)) if want_S else None
        g = np.zeros(n)
        h.check(h.L.apexgpu_get_schur(h.h, capi.ptr(S), capi.ptr(g)))
        return S, g

    def schur_matvec(self, lam: float, x: np.ndarray, explicit=True, implicit=True):
        """y = S x (reference camera-side column order) through the explicit tiles and the matrix-free operator."""
        h = self._need()
        x = np.ascontiguousarray(x, dtype=np.float64)
        ye = np.zeros_like(x) if explicit else None
        yi = np.zeros_like(x) if implicit else None
        h.check(h.L.apexgpu_schur_matvec(h.h, float(lam), capi.ptr(x), capi.ptr(ye), capi.ptr(yi)))
        return ye, yi

    def get_hessian(self):
        """LinearSolver::get_hessian: H = J^T J (undamped) as a scipy.sparse.csc_matrix in the global column order."""
        import scipy.sparse as sp

        h = self._need()
        nnz = C.c_int64(0)
        h.check(h.L.apexgpu_get_hessian_csc(h.h, C.byref(nnz), None, None, None))
        n = self._problem.total_dof
        colptr = np.zeros(n + 1, dtype=np.int64); rowidx = np.zeros(nnz.value, dtype=np.int64); vals = np.zeros(nnz.value)
        h.check(h.L.apexgpu_get_hessian_csc(h.h, C.byref(nnz), capi.ptr(colptr), capi.ptr(rowidx), capi.ptr(vals)))
        return sp.csc_matrix((vals, rowidx, colptr), shape=(n, n))

    def get_landmark_blocks(self):
        h = self._need()
        hi = np.zeros((h.n_pt, 3, 3)); gl = np.zeros((h.n_pt, 3))
        h.check(h.L.apexgpu_get_landmark_blocks(h.h, capi.ptr(hi), capi.ptr(gl)))
        return hi, gl

    def info(self) -> dict:
        h = self._need(); out = (C.c_double * 16)()
        h.check(h.L.apexgpu_info(h.h, C.byref(out)))
        return dict(tile_rows=int(out[0]), tiles=int(out[1]), pair_blocks=float(out[2]), cam_dof=int(out[3]),
                    last_reg=float(out[4]), pcg_iterations=int(out[5]), touched_tiles=int(out[6]), local_obs=int(out[7]), etree_levels=int(out[8]),
                    n_potrf=int(out[9]), n_trsm=int(out[10]), n_update=int(out[11]),
                    dist_top_columns=int(out[12]), dist_local_fraction=float(out[13]), tree_sharded=bool(out[14]), schur_form=int(out[15]))

    def variant_info(self, asked: "SchurVariant | None" = None) -> dict:
        """Which variant a solve of `asked` (default: this solver's variant) really runs on this handle, and why: after the
        automatic selection at initialize_structure (a structure whose direct factorisation was refused) every variant is
        answered by the matrix-free PCG."""
        h = self._need()
        asked = self.variant if asked is None else asked
        used = C.c_int(0); buf = C.create_string_buffer(512)
        h.check(h.L.apexgpu_variant_info(h.h, asked.value, C.byref(used), buf, 512))
        c = (C.c_double * 4)()
        h.check(h.L.apexgpu_variant_costs(h.h, C.byref(c)))
        return dict(variant_asked=asked.name, variant_used=SchurVariant(used.value).name, reason=buf.value.decode(),
                    predicted_direct_ms=float(c[0]), predicted_matrix_free_ms=float(c[1]),
                    variant_choice=("direct", "matrix-free by predicted cost", "matrix-free: plan refused", "matrix-free by option")[int(c[2])])

    def counters(self) -> dict:
        """Events of this handle's life: dataflow triangular sweeps that timed out and were repeated level by level."""
        h = self._need(); out = (C.c_int64 * 4)()
        h.check(h.L.apexgpu_counters(h.h, C.byref(out)))
        return dict(sweep_timeouts=int(out[0]), tri_dataflow=bool(out[1]), factor_flow_timeouts=int(out[2]), factor_flow_groups=int(out[3]))

    def setup_times(self) -> dict:
        """Wall time of initialize_structure by phase (seconds) and the counts that go with it."""
        h = self._need(); sec = (C.c_double * 6)(); cnt = (C.c_double * 4)()
        h.check(h.L.apexgpu_setup_times(h.h, C.byref(sec), C.byref(cnt)))
        return dict(order=sec[0], lists=sec[1], tile_plan=sec[2], schur_lists=sec[3], uploads=sec[4], total=sec[5],
                    hub_cameras=int(cnt[0]), pair_blocks=int(cnt[1]), pair_slots=int(cnt[2]), schur_form=int(cnt[3]))

    def set_option(self, name: str, value: int):
        h = self._need(); h.check(h.L.apexgpu_set_option(h.h, name.encode(), int(value)))

    def pair_records(self) -> np.ndarray:
        """Tests: the records of the sorted pair list as they sit on the device, (slots, 4) uint32 = (i, j, landmark, queue)."""
        h = self._need()
        n = self.setup_times()["pair_slots"]
        out = np.zeros((n, 4), dtype=np.uint32)
        h.check(h.L.apexgpu_debug_get_pair_records(h.h, capi.ptr(out), n))
        return out

    def owned_landmarks(self) -> np.ndarray:
        """Boolean mask (caller's landmark numbering) of the landmarks this rank assembles and back-substitutes."""
        h = self._need()
        m = np.zeros(h.n_pt, dtype=np.uint8)
        h.check(h.L.apexgpu_owned_landmarks(h.h, capi.ptr(m)))
        return m.astype(bool)

    def export_step(self):
        """Step and gradient of the last solve on this handle (global column order)."""
        h = self._need()
        step = np.zeros(self._problem.total_dof); grad = np.zeros(self._problem.total_dof)
        h.check(h.L.apexgpu_export_step(h.h, capi.ptr(step), capi.ptr(grad)))
        return step, grad

    @staticmethod
    def lockstep_solve(solvers, lam: float):
        """Test helper: `solvers` are the ranks 0..n-1 of one sharded problem in this process (with_shard(r, n));
        runs one distributed Cholesky solve in lockstep, the library playing the all-reduces (apexgpu.h)."""
        hs = (C.c_void_p * len(solvers))(*[s._need().h for s in solvers])
        h0 = solvers[0]._need()
        rc = h0.L.apexgpu_debug_lockstep_solve(hs, len(solvers), float(lam))
        if rc != 0:
            for s in solvers:
                msg = s._need().L.apexgpu_last_error(s._need().h)
                if msg:
                    raise capi.LinAlgError(rc, msg.decode())
            raise capi.LinAlgError(rc, "lockstep solve")

    def enable_stage_timing(self, on=True): h = self._need(); h.check(h.L.apexgpu_enable_stage_timing(h.h, int(on)))
    def reset_stage_times(self): h = self._need(); h.check(h.L.apexgpu_reset_stage_times(h.h))

    def stage_times(self) -> dict:
        h = self._need()
        ms = (C.c_double * capi.NUM_STAGES)(); n = (C.c_int64 * capi.NUM_STAGES)()
        h.check(h.L.apexgpu_stage_times(h.h, C.byref(ms), C.byref(n)))
        return {name: (ms[i], n[i]) for i, name in enumerate(capi.STAGE_NAMES)}

    def lm_optimize(self, cfg: LevenbergMarquardtConfig):
        h = self._need()
        c = cfg.to_c()
        c.variant = self.variant.value
        res = capi.LmResultC()
        cap = cfg.max_iterations + 2
        hist = (capi.LmIterC * cap)()
        h.check(h.L.apexgpu_lm_optimize(h.h, C.byref(c), C.byref(res), C.cast(hist, C.c_void_p), cap))
        n = res.iterations
        H = np.array([[getattr(hist[i], f) for f, _ in capi.LmIterC._fields_] for i in range(min(n, cap))])
        return res, H.reshape(-1, 8), c

    def close(self):
        if self._h is not None:
            self._h.close()
            self._h = None


@dataclass
class SolverResult:
    """src/optimizer/mod.rs:250-273 (BA subset)."""

    status: OptimizationStatus
    iterations: int
    initial_cost: float
    final_cost: float
    parameters: tuple  # (poses, intr, points)
    elapsed_time: float
    final_gradient_norm: float
    final_parameter_update_norm: float
    cost_evaluations: int
    jacobian_evaluations: int
    successful_steps: int
    unsuccessful_steps: int
    history: np.ndarray  # per iteration: cost, damping, rho, accepted, |g|, |step|, predicted, trial cost
    final_damping: float = 0.0


class LevenbergMarquardt:
    def __init__(self, config: LevenbergMarquardtConfig | None = None, device: int = 0):
        self.config = config or LevenbergMarquardtConfig()
        self.device = device
        self.linear_solver: GpuSchurComplementSolver | None = None

    @classmethod
    def new(cls): return cls()

    @classmethod
    def with_config(cls, config: LevenbergMarquardtConfig, device: int = 0): return cls(config, device)

    def optimize(self, problem: Problem, initial_values=None, solver: GpuSchurComplementSolver | None = None) -> SolverResult:
        """LevenbergMarquardt::optimize (:1034-1083): the GpuSchurComplement arm builds the solver,
        initialises its structure, uploads the initial values and runs the loop on the device."""
        t = self.config.linear_solver_type
        from .pose_graph import GpuSparseCholeskySolver, PoseGraphProblem
        if isinstance(problem, PoseGraphProblem):
            # the SparseCholesky arm (levenberg_marquardt.rs:1055-1062) on the device
            if t not in (LinearSolverType.GpuSparseCholesky, LinearSolverType.SparseCholesky):
                raise NotImplementedError(f"{t}: a pose graph has no landmarks to eliminate; use SparseCholesky")
            s = solver or GpuSparseCholeskySolver(self.device)
            if s._h is None:
                s.initialize_structure(problem)
            s.set_parameters(problem.data.poses if initial_values is None else initial_values)
            self.linear_solver = s
            res, hist, c = s.lm_optimize(self.config)
            return SolverResult(
                status=OptimizationStatus(res.status), iterations=res.iterations, initial_cost=res.initial_cost,
                final_cost=res.final_cost, parameters=(s.get_parameters(),), elapsed_time=res.elapsed_s,
                final_gradient_norm=res.final_gradient_norm, final_parameter_update_norm=res.final_step_norm,
                cost_evaluations=res.cost_evaluations, jacobian_evaluations=res.jacobian_evaluations,
                successful_steps=res.successful_steps, unsuccessful_steps=res.unsuccessful_steps, history=hist,
                final_damping=c.damping)
        if t not in (LinearSolverType.GpuSchurComplement, LinearSolverType.SparseSchurComplement):
            raise NotImplementedError(f"{t} is a CPU solver of the reference; this backend provides GpuSchurComplement")
        d = problem.data
        s = solver or GpuSchurComplementSolver(self.device)
        s.with_variant(self.config.schur_variant).with_preconditioner(self.config.schur_preconditioner)
        if s._h is None:
            if self.config.schur_variant == SchurVariant.Implicit and solver is None:
                # IterativeSchurSolver never forms S (implicit_schur.rs:163-251): neither does a handle made for it
                s.with_option("matrix_free_only", 1)
            s.initialize_structure(problem)
        poses, intr, pts = initial_values if initial_values is not None else (d.poses, d.intr, d.points)
        s.set_parameters(poses, intr, pts)
        self.linear_solver = s
        res, hist, c = s.lm_optimize(self.config)
        return SolverResult(
            status=OptimizationStatus(res.status), iterations=res.iterations, initial_cost=res.initial_cost,
            final_cost=res.final_cost, parameters=s.get_parameters(), elapsed_time=res.elapsed_s,
            final_gradient_norm=res.final_gradient_norm, final_parameter_update_norm=res.final_step_norm,
            cost_evaluations=res.cost_evaluations, jacobian_evaluations=res.jacobian_evaluations,
            successful_steps=res.successful_steps, unsuccessful_steps=res.unsuccessful_steps, history=hist,
            final_damping=c.damping)
