"""SE3 pose-graph path (BASELINE.json configs[1]): Python mirror of what bin/pose_graph_g2o.rs drives --
`G2oLoader` (crates/apex-io/src/g2o.rs) over the library's C++ reader, the `Problem` of BetweenFactor<SE3>
blocks with the first vertex fixed, and the `SparseCholesky` linear solver on the device
(`GpuSparseCholeskySolver`, src/linalg/sparse/cholesky.rs:159-230).  Every numeric path calls
libapexgpu.so; there is no CPU fallback."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import capi
from .synthetic import PoseGraphData

G2O_ERROR_NAMES = {-30: "Io", -31: "Parse", -32: "MissingFields", -33: "InvalidNumber", -34: "DuplicateVertex",
                   -35: "InvalidQuaternion"}


class G2oError(RuntimeError):
    """Mirror of apex_io::IoError for the G2O reader: `.kind` is the variant name."""

    def __init__(self, code: int, message: str):
        self.code = code
        self.kind = G2O_ERROR_NAMES.get(code, f"Error({code})")
        super().__init__(f"{self.kind}: {message}")


@dataclass
class G2oGraph:
    """apex_io::Graph restricted to SE3 (crates/apex-io/src/lib.rs:336-341), columnar, file order."""

    vertex_ids: np.ndarray      # (n_v,) int64
    vertex_poses: np.ndarray    # (n_v, 7) [t, qw,qx,qy,qz]
    edge_from: np.ndarray       # (n_e,) int64 vertex ids
    edge_to: np.ndarray
    edge_measurements: np.ndarray  # (n_e, 7)
    edge_information: np.ndarray   # (n_e, 6, 6)
    n_vertices_se2: int = 0
    n_edges_se2: int = 0
    _problem: PoseGraphData | None = field(default=None, repr=False)

    def vertex_count(self) -> int:
        return int(self.vertex_ids.shape[0]) + self.n_vertices_se2

    def edge_count(self) -> int:
        return int(self.edge_from.shape[0]) + self.n_edges_se2

    def to_problem_data(self, name: str = "g2o") -> PoseGraphData:
        assert self._problem is not None
        p = self._problem
        return PoseGraphData(ids=p.ids.copy(), poses=p.poses.copy(), e_from=p.e_from.copy(), e_to=p.e_to.copy(),
                             meas=p.meas.copy(), name=name)


class G2oLoader:
    @staticmethod
    def load(path) -> G2oGraph:
        L = capi.load()
        h = C.c_void_p()
        rc = L.apexgpu_g2o_open(str(path).encode(), C.byref(h))
        if rc != 0:
            raise G2oError(rc, L.apexgpu_g2o_last_error().decode())
        try:
            nv, ne, nv2, ne2 = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
            L.apexgpu_g2o_sizes(h, C.byref(nv), C.byref(ne), C.byref(nv2), C.byref(ne2))
            nv, ne = nv.value, ne.value
            ids = np.zeros(nv, np.int64); poses = np.zeros((nv, 7)); ef = np.zeros(ne, np.int64); et = np.zeros(ne, np.int64)
            meas = np.zeros((ne, 7)); info = np.zeros((ne, 6, 6))
            L.apexgpu_g2o_raw(h, capi.ptr(ids), capi.ptr(poses), capi.ptr(ef), capi.ptr(et), capi.ptr(meas), capi.ptr(info))
            sid = np.zeros(nv, np.int64); sp = np.zeros((nv, 7)); pf = np.zeros(ne, np.uint32); pt = np.zeros(ne, np.uint32)
            pm = np.zeros((ne, 7))
            rc = L.apexgpu_g2o_problem(h, capi.ptr(sid), capi.ptr(sp), capi.ptr(pf), capi.ptr(pt), capi.ptr(pm), None, None)
            if rc != 0:
                raise G2oError(rc, L.apexgpu_g2o_last_error().decode())
            prob = PoseGraphData(ids=sid, poses=sp, e_from=pf, e_to=pt, meas=pm)
            return G2oGraph(ids, poses, ef, et, meas, info, nv2.value, ne2.value, prob)
        finally:
            L.apexgpu_g2o_close(h)


def write_g2o(path, data: PoseGraphData, information: np.ndarray | None = None):
    """G2oLoader::write for SE3 graphs (g2o.rs:20-135): vertices sorted by id, `{:.17e}` numbers,
    21 upper-triangular information values per edge (identity unless given)."""
    def fmt(x):
        return f"{float(x):.17e}"
    with open(path, "w") as f:
        f.write("# G2O file written by Apex Solver\n")
        f.write(f"# SE2 vertices: 0, SE3 vertices: {data.n_v}, SE2 edges: 0, SE3 edges: {data.n_e}\n\n")
        order = np.argsort(data.ids, kind="stable")
        for k in order:
            p = data.poses[k]
            f.write("VERTEX_SE3:QUAT %d %s\n" % (data.ids[k], " ".join(fmt(v) for v in (p[0], p[1], p[2], p[4], p[5], p[6], p[3]))))
        for e in range(data.n_e):
            m = data.meas[e]
            I = np.eye(6) if information is None else information[e]
            iu = [I[i, j] for i in range(6) for j in range(i, 6)]
            f.write("EDGE_SE3:QUAT %d %d %s %s\n" % (data.ids[data.e_from[e]], data.ids[data.e_to[e]],
                    " ".join(fmt(v) for v in (m[0], m[1], m[2], m[4], m[5], m[6], m[3])), " ".join(fmt(v) for v in iu)))


def pose_graph_columns(ids: np.ndarray) -> np.ndarray:
    """First global column of `x{id}` in sorted-name order (src/optimizer/mod.rs:530-536)."""
    L = capi.load()
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    out = np.zeros(ids.shape[0], np.int64)
    rc = L.apexgpu_pose_graph_columns(ids.shape[0], capi.ptr(ids), capi.ptr(out))
    if rc != 0:
        raise G2oError(rc, L.apexgpu_g2o_last_error().decode())
    return out


def se3_as_vector(pose7) -> np.ndarray:
    """SE3::from(DVector).to_vector(): [t, w, i, j, k] with the quaternion normalised (se3.rs:200-206, 107-113)."""
    p = np.asarray(pose7, dtype=np.float64).reshape(7).copy()
    q = p[3:7]
    q = q / np.sqrt(np.dot(q, q))
    q = q / np.sqrt(np.dot(q, q))
    p[3:7] = q
    return p


@dataclass
class PoseGraphProblem:
    """The factor graph bin/pose_graph_g2o.rs:748-830 builds: variables `x{id}` (SE3), one
    BetweenFactor(measurement) per edge on (x{from}, x{to}), optional loss on every block."""

    data: PoseGraphData
    huber_delta: float | None = None
    fix: np.ndarray = field(default=None)
    priors: list = field(default_factory=list)   # (vertex index, data[7], huber delta or None) per PriorFactor block

    def __post_init__(self):
        if self.fix is None:
            self.fix = np.zeros((self.data.n_v, 6), dtype=np.uint8)
        self.pose_col = pose_graph_columns(self.data.ids)

    def add_prior(self, name: str, data=None, huber_delta: float | None = None):
        """`problem.add_residual_block(&[name], PriorFactor { data }, loss)` (src/factors/prior_factor.rs:53-113): the
        gauge of the reference's pose-graph integration test (tests/integration_tests.rs:98-118, Huber(1.0), data = the
        variable's initial value).  r = to_vector(x) - data, seven rows; the linearizer keeps the first six columns of the
        7 x 7 identity Jacobian (src/linearizer/cpu/sparse.rs:201-204)."""
        if not name.startswith("x"):
            raise KeyError(name)
        hit = np.nonzero(self.data.ids == int(name[1:]))[0]
        if hit.size == 0:
            raise KeyError(name)
        v = int(hit[0])
        x = se3_as_vector(self.data.poses[v]) if data is None else np.asarray(data, dtype=np.float64).reshape(7)
        self.priors.append((v, x, huber_delta))
        return self

    @classmethod
    def pose_graph(cls, data: PoseGraphData, huber_delta: float | None = None) -> "PoseGraphProblem":
        """The LM set-up: all six DOF of the first vertex fixed (pose_graph_g2o.rs:790-797)."""
        p = cls(data, huber_delta)
        for dof in range(6):
            p.fix_variable(f"x{int(data.ids[0])}", dof)
        return p

    def fix_variable(self, name: str, dof: int):
        """Problem::fix_variable (src/core/problem.rs:609-616)."""
        if not name.startswith("x"):
            raise KeyError(name)
        hit = np.nonzero(self.data.ids == int(name[1:]))[0]
        if hit.size == 0:
            raise KeyError(name)
        self.fix[hit[0], dof] = 1

    @property
    def total_dof(self) -> int:
        return 6 * self.data.n_v

    @property
    def num_residual_blocks(self) -> int:
        return self.data.n_e + len(self.priors)


class GpuSparseCholeskySolver:
    """Device counterpart of SparseCholeskySolver (src/linalg/sparse/cholesky.rs) for pose graphs:
    `solve_augmented_equation(lambda)` linearises the BetweenFactors, assembles J^T J + lambda I
    block-sparse and solves by the tile Cholesky.  No Schur complement."""

    def __init__(self, device: int = 0):
        self.device = device
        self._h: capi.PgHandle | None = None
        self._opts: dict[str, int] = {}
        self.problem: PoseGraphProblem | None = None

    def with_option(self, name: str, value: int):
        self._opts[name] = int(value)
        return self

    def initialize_structure(self, problem: PoseGraphProblem):
        d = problem.data
        self.close()
        h = capi.PgHandle(d.n_v, d.n_e, self.device)
        for k, v in self._opts.items():
            h.check(h.L.apexgpu_pg_set_option(h.h, k.encode(), v))
        ef = np.ascontiguousarray(d.e_from, dtype=np.uint32); et = np.ascontiguousarray(d.e_to, dtype=np.uint32)
        meas = np.ascontiguousarray(d.meas, dtype=np.float64)
        fix = np.ascontiguousarray(problem.fix, dtype=np.uint8)
        col = np.ascontiguousarray(problem.pose_col, dtype=np.int64)
        delta = -1.0 if problem.huber_delta is None else float(problem.huber_delta)
        h.check(h.L.apexgpu_pg_set_structure(h.h, capi.ptr(ef), capi.ptr(et), capi.ptr(meas), capi.ptr(col), capi.ptr(fix), delta))
        self._h, self.problem = h, problem
        if problem.priors:
            self.set_priors(problem.priors)
        return self

    def set_priors(self, priors):
        """PriorFactor blocks: (vertex index, data[7], huber delta or None) each; replaces the set."""
        h = self._need()
        v = np.ascontiguousarray([q[0] for q in priors], dtype=np.uint32)
        x = np.ascontiguousarray([q[1] for q in priors], dtype=np.float64).reshape(-1, 7)
        dl = np.ascontiguousarray([-1.0 if q[2] is None else float(q[2]) for q in priors], dtype=np.float64)
        h.check(h.L.apexgpu_pg_set_priors(h.h, len(v), capi.ptr(v), capi.ptr(x), capi.ptr(dl)))

    def get_prior_residual(self) -> np.ndarray:
        """Corrected residuals of the prior blocks at the current parameters: [n_prior][7]."""
        h = self._need()
        n = len(self.problem.priors)
        r = np.zeros((n, 7))
        if n: h.check(h.L.apexgpu_pg_get_prior_residual(h.h, capi.ptr(r)))
        return r

    def _need(self) -> capi.PgHandle:
        if self._h is None:
            raise capi.LinAlgError(-6, "Block structure not built. Call initialize_structure() first.")
        return self._h

    def set_parameters(self, poses):
        h = self._need()
        p = np.ascontiguousarray(poses, dtype=np.float64)
        h.check(h.L.apexgpu_pg_set_params(h.h, capi.ptr(p)))

    def get_parameters(self) -> np.ndarray:
        h = self._need()
        p = np.zeros((h.n_vertices, 7))
        h.check(h.L.apexgpu_pg_get_params(h.h, capi.ptr(p)))
        return p

    def compute_cost(self) -> float:
        h = self._need()
        c = C.c_double()
        h.check(h.L.apexgpu_pg_cost(h.h, C.byref(c)))
        return c.value

    def solve_augmented_equation(self, lam: float, want_step: bool = True):
        h = self._need()
        n = 6 * h.n_vertices
        step = np.zeros(n) if want_step else None
        self._grad = np.zeros(n) if want_step else None
        h.check(h.L.apexgpu_pg_solve_augmented(h.h, float(lam), capi.ptr(step), capi.ptr(self._grad)))
        return step

    def solve_normal_equation(self):
        return self.solve_augmented_equation(0.0)

    def get_gradient(self):
        return getattr(self, "_grad", None)

    # AssemblyBackend::compute_column_norms / apply_column_scaling / apply_inverse_scaling (linearizer/mod.rs:229-262)
    def compute_column_norms(self) -> np.ndarray:
        h = self._need()
        n = np.zeros(6 * h.n_vertices)
        h.check(h.L.apexgpu_pg_column_norms(h.h, capi.ptr(n)))
        return n

    def apply_column_scaling(self, scaling):
        h = self._need()
        a = None if scaling is None else np.ascontiguousarray(scaling, dtype=np.float64)
        if a is not None and a.shape != (6 * h.n_vertices,):
            raise ValueError("scaling must have total_dof entries")
        h.check(h.L.apexgpu_pg_set_column_scaling(h.h, capi.ptr(a)))
        self._scaling = a

    def apply_inverse_scaling(self, step):
        s = getattr(self, "_scaling", None)
        return step if s is None else step * s

    def step_stats(self):
        h = self._need()
        o = (C.c_double * 3)()
        h.check(h.L.apexgpu_pg_step_stats(h.h, C.byref(o)))
        return o[0], o[1], o[2]

    def eval_step(self) -> float:
        h = self._need()
        c = C.c_double()
        h.check(h.L.apexgpu_pg_eval_step(h.h, C.byref(c)))
        return c.value

    def commit_step(self): h = self._need(); h.check(h.L.apexgpu_pg_commit_step(h.h))
    def discard_step(self): h = self._need(); h.check(h.L.apexgpu_pg_discard_step(h.h))

    def parameter_norm(self) -> float:
        h = self._need()
        c = C.c_double()
        h.check(h.L.apexgpu_pg_parameter_norm(h.h, C.byref(c)))
        return c.value

    def get_residual(self) -> np.ndarray:
        h = self._need()
        r = np.zeros((h.n_edges, 6))
        h.check(h.L.apexgpu_pg_get_residual(h.h, capi.ptr(r)))
        return r

    def get_jacobian_blocks(self) -> np.ndarray:
        h = self._need()
        j = np.zeros((h.n_edges, 6, 12))
        h.check(h.L.apexgpu_pg_get_jacobian_blocks(h.h, capi.ptr(j)))
        return j

    def get_hessian(self, lam: float = 0.0):
        h = self._need()
        n = 6 * h.n_vertices
        H = np.zeros((n, n)); g = np.zeros(n)
        h.check(h.L.apexgpu_pg_get_hessian(h.h, float(lam), capi.ptr(H), capi.ptr(g)))
        return H, g

    def info(self) -> dict:
        h = self._need()
        a = (C.c_double * 8)()
        h.check(h.L.apexgpu_pg_info(h.h, C.byref(a)))
        return {"tile_rows": int(a[0]), "tiles": int(a[1]), "touched_tiles": int(a[2]), "etree_levels": int(a[3]),
                "total_dof": int(a[4]), "n_potrf": int(a[5]), "n_trsm": int(a[6]), "n_update": int(a[7])}

    def set_option(self, name: str, value: int):
        h = self._need(); h.check(h.L.apexgpu_pg_set_option(h.h, name.encode(), int(value)))

    def counters(self) -> dict:
        h = self._need(); out = (C.c_int64 * 4)()
        h.check(h.L.apexgpu_pg_counters(h.h, C.byref(out)))
        return dict(sweep_timeouts=int(out[0]), tri_dataflow=bool(out[1]), factor_flow_timeouts=int(out[2]), factor_flow_groups=int(out[3]))

    def enable_stage_timing(self, on=True): h = self._need(); h.check(h.L.apexgpu_pg_enable_stage_timing(h.h, int(on)))
    def reset_stage_times(self): h = self._need(); h.check(h.L.apexgpu_pg_reset_stage_times(h.h))

    def stage_times(self) -> dict:
        h = self._need()
        ms = (C.c_double * capi.PG_NUM_STAGES)(); n = (C.c_int64 * capi.PG_NUM_STAGES)()
        h.check(h.L.apexgpu_pg_stage_times(h.h, C.byref(ms), C.byref(n)))
        return {name: (ms[i], n[i]) for i, name in enumerate(capi.PG_STAGE_NAMES)}

    def lm_optimize(self, cfg):
        h = self._need()
        c = cfg.to_c()
        c.variant = 0
        res = capi.LmResultC()
        cap = cfg.max_iterations + 2
        hist = (capi.LmIterC * cap)()
        h.check(h.L.apexgpu_pg_lm_optimize(h.h, C.byref(c), C.byref(res), C.cast(hist, C.c_void_p), cap))
        n = res.iterations
        H = np.array([[getattr(hist[i], f) for f, _ in capi.LmIterC._fields_] for i in range(min(n, cap))])
        return res, H.reshape(-1, 8), c

    def close(self):
        if self._h is not None:
            self._h.close()
            self._h = None
