"""ctypes binding of the C-ABI library (include/apexgpu.h).

The library is the product: there is no Python/CPU fallback.  Loading fails loudly when
`libapexgpu.so` is missing; every compute entry point returns an error when no MI355X is visible.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libapexgpu.so")
LIB_PATH = os.environ.get("APEXGPU_LIB", LIB_PATH)   # (A/B of two builds on one box: tools only)

NUM_STAGES = 10
STAGE_NAMES = ("cam_reduce", "landmark_reduce", "schur_scatter", "all_reduce", "factor", "tri_solve",
               "back_substitute", "step_stats", "retract", "cost")

# every symbol include/apexgpu.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = (
    "apexgpu_create", "apexgpu_destroy", "apexgpu_last_error", "apexgpu_version", "apexgpu_set_structure",
    "apexgpu_set_cg_params", "apexgpu_set_params", "apexgpu_get_params", "apexgpu_cost", "apexgpu_assemble", "apexgpu_solve_augmented",
    "apexgpu_step_stats", "apexgpu_eval_step", "apexgpu_commit_step", "apexgpu_discard_step",
    "apexgpu_parameter_norm", "apexgpu_column_norms", "apexgpu_set_column_scaling", "apexgpu_lm_optimize", "apexgpu_get_residual", "apexgpu_get_jacobian_blocks",
    "apexgpu_get_schur", "apexgpu_get_landmark_blocks", "apexgpu_get_hessian_csc", "apexgpu_debug_invert_blocks", "apexgpu_debug_pair_lists", "apexgpu_debug_pair_lists_queued", "apexgpu_debug_pair_lists_queued_dc", "apexgpu_debug_host_structure", "apexgpu_setup_times", "apexgpu_schur_matvec", "apexgpu_set_option", "apexgpu_enable_stage_timing", "apexgpu_reset_stage_times",
    "apexgpu_stage_times", "apexgpu_info", "apexgpu_variant_info", "apexgpu_variant_costs", "apexgpu_trim_host_cache", "apexgpu_host_cache_bytes", "apexgpu_counters", "apexgpu_debug_get_pair_records", "apexgpu_get_unique_id", "apexgpu_comm_init", "apexgpu_comm_init_shm", "apexgpu_set_shard", "apexgpu_shard_range",
    "apexgpu_debug_lockstep_solve", "apexgpu_export_step", "apexgpu_owned_landmarks", "apexgpu_debug_partition", "apexgpu_debug_check_schedule",
    "apexgpu_bal_open", "apexgpu_bal_close", "apexgpu_bal_last_error", "apexgpu_bal_sizes", "apexgpu_bal_raw",
    "apexgpu_bal_variables", "apexgpu_reference_columns",
    # SE3 pose-graph backend
    "apexgpu_pg_create", "apexgpu_pg_destroy", "apexgpu_pg_last_error", "apexgpu_pg_set_structure", "apexgpu_pg_set_params",
    "apexgpu_pg_get_params", "apexgpu_pg_cost", "apexgpu_pg_solve_augmented", "apexgpu_pg_step_stats", "apexgpu_pg_eval_step",
    "apexgpu_pg_commit_step", "apexgpu_pg_discard_step", "apexgpu_pg_parameter_norm", "apexgpu_pg_column_norms",
    "apexgpu_pg_set_column_scaling", "apexgpu_pg_lm_optimize",
    "apexgpu_pg_get_residual", "apexgpu_pg_get_jacobian_blocks", "apexgpu_pg_get_hessian", "apexgpu_pg_set_option",
    "apexgpu_pg_enable_stage_timing", "apexgpu_pg_reset_stage_times", "apexgpu_pg_stage_times", "apexgpu_pg_info", "apexgpu_pg_counters",
    "apexgpu_pg_set_priors", "apexgpu_pg_get_prior_residual",
    "apexgpu_g2o_open", "apexgpu_g2o_close", "apexgpu_g2o_last_error", "apexgpu_g2o_sizes", "apexgpu_g2o_raw",
    "apexgpu_g2o_problem", "apexgpu_pose_graph_columns",
)
PG_NUM_STAGES = 6
PG_STAGE_NAMES = ("assemble", "factor", "tri_solve", "step_stats", "retract", "cost")
_NON_INT = ("apexgpu_destroy", "apexgpu_last_error", "apexgpu_version", "apexgpu_host_cache_bytes", "apexgpu_bal_close", "apexgpu_bal_last_error",
            "apexgpu_pg_destroy", "apexgpu_pg_last_error", "apexgpu_g2o_close", "apexgpu_g2o_last_error")

ERROR_NAMES = {
    -1: "FactorizationFailed", -2: "SingularMatrix", -3: "SparseMatrixCreation", -4: "MatrixConversion",
    -5: "InvalidInput", -6: "InvalidState", -10: "DeviceError",
}


class LinAlgError(RuntimeError):
    """Mirror of apex-solver's LinAlgError (src/linalg/mod.rs:76-101): `.kind` is the variant name."""

    def __init__(self, code: int, message: str):
        self.code = code
        self.kind = ERROR_NAMES.get(code, f"Error({code})")
        super().__init__(f"{self.kind}: {message}")


class LmConfigC(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int), ("cost_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
        ("gradient_tolerance", C.c_double), ("damping", C.c_double), ("damping_min", C.c_double),
        ("damping_max", C.c_double), ("damping_nu", C.c_double), ("trust_region_radius", C.c_double),
        ("min_trust_region_radius", C.c_double), ("min_cost_threshold", C.c_double), ("timeout_s", C.c_double),
        ("variant", C.c_int), ("use_jacobi_scaling", C.c_int),
    ]


class LmIterC(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("cost", "damping", "rho", "accepted", "gradient_norm", "step_norm",
                                          "predicted_reduction", "trial_cost")]


class LmResultC(C.Structure):
    _fields_ = [
        ("status", C.c_int), ("iterations", C.c_int), ("initial_cost", C.c_double), ("final_cost", C.c_double),
        ("final_gradient_norm", C.c_double), ("final_step_norm", C.c_double), ("elapsed_s", C.c_double),
        ("cost_evaluations", C.c_int), ("jacobian_evaluations", C.c_int), ("successful_steps", C.c_int),
        ("unsuccessful_steps", C.c_int),
    ]


_lib = None


def load() -> C.CDLL:
    """Load libapexgpu.so (built by __graft_entry__.build() / `make -C apex-solver_amd/csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()')."
            " There is no CPU fallback."
        )
    L = C.CDLL(LIB_PATH)
    vp, i64, dbl = C.c_void_p, C.c_int64, C.c_double
    L.apexgpu_create.argtypes = [i64, i64, i64, C.c_int, C.c_int, C.POINTER(vp)]
    L.apexgpu_destroy.argtypes = [vp]
    L.apexgpu_destroy.restype = None
    L.apexgpu_last_error.argtypes = [vp]
    L.apexgpu_last_error.restype = C.c_char_p
    L.apexgpu_version.restype = C.c_char_p
    L.apexgpu_set_structure.argtypes = [vp] + [vp] * 9 + [dbl]
    L.apexgpu_set_cg_params.argtypes = [vp, C.c_int, dbl]
    L.apexgpu_set_params.argtypes = [vp, vp, vp, vp]
    L.apexgpu_get_params.argtypes = [vp, vp, vp, vp]
    L.apexgpu_cost.argtypes = [vp, C.POINTER(dbl)]
    L.apexgpu_assemble.argtypes = [vp, dbl]
    L.apexgpu_solve_augmented.argtypes = [vp, dbl, C.c_int, vp, vp]
    L.apexgpu_step_stats.argtypes = [vp, C.POINTER(dbl * 3)]
    L.apexgpu_eval_step.argtypes = [vp, C.POINTER(dbl)]
    L.apexgpu_commit_step.argtypes = [vp]
    L.apexgpu_discard_step.argtypes = [vp]
    L.apexgpu_parameter_norm.argtypes = [vp, C.POINTER(dbl)]
    L.apexgpu_column_norms.argtypes = [vp, vp]
    L.apexgpu_debug_lockstep_solve.argtypes = [vp, C.c_int, C.c_double]
    L.apexgpu_export_step.argtypes = [vp, vp, vp]
    L.apexgpu_owned_landmarks.argtypes = [vp, vp]
    L.apexgpu_debug_partition.argtypes = [C.c_int, vp, C.c_int, vp]
    L.apexgpu_debug_check_schedule.argtypes = [C.c_int, vp, C.c_int, C.c_int, vp, vp, C.c_char_p, C.c_int]
    L.apexgpu_set_column_scaling.argtypes = [vp, vp]
    L.apexgpu_lm_optimize.argtypes = [vp, C.POINTER(LmConfigC), C.POINTER(LmResultC), vp, C.c_int]
    L.apexgpu_get_residual.argtypes = [vp, vp]
    L.apexgpu_get_jacobian_blocks.argtypes = [vp, vp, vp]
    L.apexgpu_get_schur.argtypes = [vp, vp, vp]
    L.apexgpu_get_landmark_blocks.argtypes = [vp, vp, vp]
    L.apexgpu_schur_matvec.argtypes = [vp, dbl, vp, vp, vp]
    L.apexgpu_set_option.argtypes = [vp, C.c_char_p, C.c_int]
    L.apexgpu_get_hessian_csc.argtypes = [vp, vp, vp, vp, vp]
    L.apexgpu_debug_invert_blocks.argtypes = [C.c_int, C.c_int64, vp, vp, vp]
    L.apexgpu_debug_host_structure.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]
    L.apexgpu_setup_times.argtypes = [vp, vp, vp]
    L.apexgpu_debug_pair_lists.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]
    L.apexgpu_debug_pair_lists_queued.argtypes = [C.c_int64, C.c_int64, C.c_int64, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.apexgpu_enable_stage_timing.argtypes = [vp, C.c_int]
    L.apexgpu_reset_stage_times.argtypes = [vp]
    L.apexgpu_stage_times.argtypes = [vp, C.POINTER(dbl * NUM_STAGES), C.POINTER(i64 * NUM_STAGES)]
    L.apexgpu_info.argtypes = [vp, C.POINTER(dbl * 16)]
    L.apexgpu_trim_host_cache.argtypes = [C.POINTER(C.c_int64)]
    L.apexgpu_host_cache_bytes.argtypes = []
    L.apexgpu_host_cache_bytes.restype = C.c_int64
    L.apexgpu_debug_get_pair_records.argtypes = [vp, vp, i64]
    L.apexgpu_variant_info.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.c_char_p, C.c_int]
    L.apexgpu_variant_costs.argtypes = [vp, C.POINTER(dbl * 4)]
    L.apexgpu_counters.argtypes = [vp, C.POINTER(i64 * 4)]
    L.apexgpu_get_unique_id.argtypes = [vp]
    L.apexgpu_comm_init.argtypes = [vp, C.c_int, C.c_int, vp]
    L.apexgpu_comm_init_shm.argtypes = [vp, C.c_int, C.c_int, C.c_char_p]
    L.apexgpu_set_shard.argtypes = [vp, C.c_int, C.c_int]
    L.apexgpu_shard_range.argtypes = [i64, i64, vp, C.c_int, C.c_int, C.POINTER(i64), C.POINTER(i64)]
    L.apexgpu_bal_open.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.apexgpu_bal_close.argtypes = [vp]
    L.apexgpu_bal_close.restype = None
    L.apexgpu_bal_last_error.restype = C.c_char_p
    L.apexgpu_bal_sizes.argtypes = [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]
    L.apexgpu_bal_raw.argtypes = [vp, vp, vp, vp, vp, vp]
    L.apexgpu_bal_variables.argtypes = [vp, vp, vp]
    L.apexgpu_reference_columns.argtypes = [i64, i64, vp, vp, vp]
    L.apexgpu_pg_create.argtypes = [i64, i64, C.c_int, C.POINTER(vp)]
    L.apexgpu_pg_destroy.argtypes = [vp]
    L.apexgpu_pg_destroy.restype = None
    L.apexgpu_pg_last_error.argtypes = [vp]
    L.apexgpu_pg_last_error.restype = C.c_char_p
    L.apexgpu_pg_set_structure.argtypes = [vp, vp, vp, vp, vp, vp, dbl]
    L.apexgpu_pg_set_params.argtypes = [vp, vp]
    L.apexgpu_pg_set_priors.argtypes = [vp, C.c_int64, vp, vp, vp]
    L.apexgpu_pg_get_prior_residual.argtypes = [vp, vp]
    L.apexgpu_pg_get_params.argtypes = [vp, vp]
    L.apexgpu_pg_cost.argtypes = [vp, C.POINTER(dbl)]
    L.apexgpu_pg_solve_augmented.argtypes = [vp, dbl, vp, vp]
    L.apexgpu_pg_step_stats.argtypes = [vp, C.POINTER(dbl * 3)]
    L.apexgpu_pg_eval_step.argtypes = [vp, C.POINTER(dbl)]
    L.apexgpu_pg_commit_step.argtypes = [vp]
    L.apexgpu_pg_discard_step.argtypes = [vp]
    L.apexgpu_pg_parameter_norm.argtypes = [vp, C.POINTER(dbl)]
    L.apexgpu_pg_column_norms.argtypes = [vp, vp]
    L.apexgpu_pg_set_column_scaling.argtypes = [vp, vp]
    L.apexgpu_pg_lm_optimize.argtypes = [vp, C.POINTER(LmConfigC), C.POINTER(LmResultC), vp, C.c_int]
    L.apexgpu_pg_get_residual.argtypes = [vp, vp]
    L.apexgpu_pg_get_jacobian_blocks.argtypes = [vp, vp]
    L.apexgpu_pg_get_hessian.argtypes = [vp, dbl, vp, vp]
    L.apexgpu_pg_set_option.argtypes = [vp, C.c_char_p, C.c_int]
    L.apexgpu_pg_enable_stage_timing.argtypes = [vp, C.c_int]
    L.apexgpu_pg_reset_stage_times.argtypes = [vp]
    L.apexgpu_pg_stage_times.argtypes = [vp, C.POINTER(dbl * PG_NUM_STAGES), C.POINTER(i64 * PG_NUM_STAGES)]
    L.apexgpu_pg_info.argtypes = [vp, C.POINTER(dbl * 8)]
    L.apexgpu_pg_counters.argtypes = [vp, C.POINTER(i64 * 4)]
    L.apexgpu_g2o_open.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.apexgpu_g2o_close.argtypes = [vp]
    L.apexgpu_g2o_close.restype = None
    L.apexgpu_g2o_last_error.restype = C.c_char_p
    L.apexgpu_g2o_sizes.argtypes = [vp] + [C.POINTER(i64)] * 4
    L.apexgpu_g2o_raw.argtypes = [vp] * 7
    L.apexgpu_g2o_problem.argtypes = [vp] * 8
    L.apexgpu_pose_graph_columns.argtypes = [i64, vp, vp]
    for name in SYMBOLS:
        f = getattr(L, name)
        if name not in _NON_INT:
            f.restype = C.c_int
    _lib = L
    return L


def tile_partition(present: np.ndarray, world: int):
    """Owner rank of every tile column (-1: shared top) for a distributed plan of `world` ranks; host arithmetic in the
    library, no GPU needed.  present: (nt, nt) lower-triangular 0/1 structure."""
    L = load()
    pr = np.ascontiguousarray(present, dtype=np.uint8)
    nt = pr.shape[0]
    owner = np.zeros(nt, dtype=np.int32)
    n_top = L.apexgpu_debug_partition(nt, pr.ctypes.data_as(C.c_void_p), int(world), owner.ctypes.data_as(C.c_void_p))
    if n_top < 0:
        raise LinAlgError(n_top, "apexgpu_debug_partition")
    return owner, n_top


def check_schedule(present: np.ndarray, world: int = 1, rank: int = 0, two_side: int = 1, overlap: int = 1, split_u1: int = 4,
                   flood_gate: int = 256, factor_flow: int = -1, factor_flow_rows: int = 24, old_idle_level_bug: bool = False,
                   drop_wait: int = -1) -> dict:
    """Host-only race check of the factorisation's launch sequence for one tile structure (apexgpu_debug_check_schedule)."""
    L = load()
    pr = np.ascontiguousarray(present, dtype=np.uint8)
    nt = pr.shape[0]
    opts = np.array([two_side, overlap, split_u1, flood_gate, factor_flow, factor_flow_rows, int(old_idle_level_bug), drop_wait], dtype=np.int32)
    out = np.zeros(8, dtype=np.int64)
    msg = C.create_string_buffer(512)
    rc = L.apexgpu_debug_check_schedule(nt, pr.ctypes.data_as(C.c_void_p), int(world), int(rank), opts.ctypes.data_as(C.c_void_p),
                                        out.ctypes.data_as(C.c_void_p), msg, 512)
    if rc < 0:
        raise LinAlgError(rc, "apexgpu_debug_check_schedule: " + msg.value.decode())
    return dict(levels=rc, calls=int(out[0]), launches=int(out[1]), violations=int(out[2]) + int(out[3]), violations_top=int(out[3]),
                flow_units=int(out[4]), flow_groups=int(out[5]), waits=int(out[6]), dropped=bool(out[7]), first=msg.value.decode())


HOST_STRUCTURE_STATS = ("tile_rows", "hub_cameras", "border_tiles", "touched_tiles", "tiles", "etree_levels", "top_columns",
                        "tree_sharded", "s_order", "s_lists", "s_plan", "s_schur_lists", "s_uploads", "s_total", "pair_contributions",
                        "pair_blocks")


def host_structure(n_cam: int, n_pt: int, cam_idx: np.ndarray, pt_idx: np.ndarray, mode: int = 1, rank: int = 0, world: int = 1,
                   nested_dissection: int = 1, hubs_last: int = 1, dist_factor: int = 1, tree_sharding: int = 1, schur_form: int = 3):
    """Host only: what apexgpu_set_structure derives from the observation list before it touches the device."""
    ci = np.ascontiguousarray(cam_idx, dtype=np.uint32); pi = np.ascontiguousarray(pt_idx, dtype=np.uint32)
    opts = np.array([nested_dissection, hubs_last, dist_factor, tree_sharding, schur_form], dtype=np.int32)
    stats = np.zeros(16); dc = 9 if mode in (1, 4, 5, 6) else 6
    nt = (n_cam * dc + 143) // 144
    cmap = np.zeros(n_cam, dtype=np.int32); owned = np.zeros(n_pt, dtype=np.uint8); towner = np.zeros(nt, dtype=np.int32)
    rc = load().apexgpu_debug_host_structure(n_cam, n_pt, len(ci), mode, ptr(ci), ptr(pi), rank, world, ptr(opts), ptr(stats),
                                             ptr(cmap), ptr(owned), ptr(towner))
    if rc != 0:
        raise LinAlgError(rc, "apexgpu_debug_host_structure failed")
    out = dict(zip(HOST_STRUCTURE_STATS, stats.tolist()))
    out.update(cmap=cmap, owned=owned.astype(bool), tile_owner=towner)
    return out


def pair_lists(n_cam: int, n_pt: int, dc: int, cam_idx: np.ndarray, pt_idx: np.ndarray):
    """Host only: the sorted camera-pair lists of the default Schur reduction (apexgpu_debug_pair_lists)."""
    ci = np.ascontiguousarray(cam_idx, dtype=np.uint32); pi = np.ascontiguousarray(pt_idx, dtype=np.uint32)
    counts = np.zeros(4, dtype=np.int64)
    L = load()
    rc = L.apexgpu_debug_pair_lists(n_cam, n_pt, len(ci), dc, ptr(ci), ptr(pi), ptr(counts), None, None, None, None, None)
    if rc != 0:
        raise LinAlgError(rc, "apexgpu_debug_pair_lists failed")
    recs = np.zeros((counts[0], 4), dtype=np.uint32); chunks = np.zeros((counts[1], 2), dtype=np.int32)
    blocks = np.zeros((counts[2], 4), dtype=np.int64); tasks = np.zeros((counts[3], 2), dtype=np.int32)
    o_index = np.zeros(len(ci), dtype=np.int32)
    rc = L.apexgpu_debug_pair_lists(n_cam, n_pt, len(ci), dc, ptr(ci), ptr(pi), ptr(counts), ptr(recs), ptr(chunks), ptr(blocks),
                                    ptr(tasks), ptr(o_index))
    if rc != 0:
        raise LinAlgError(rc, "apexgpu_debug_pair_lists failed")
    return dict(recs=recs, chunks=chunks, blocks=blocks, tasks=tasks, o_index=o_index)


def pair_lists_queued(n_cam: int, n_pt: int, cam_idx: np.ndarray, pt_idx: np.ndarray, dc: int = 9):
    """Host only: the same pairs in the queued layout ("schur_form" 4; apexgpu_debug_pair_lists_queued_dc): nine columns per
    camera = seven queues of nine pairs per chunk, six columns = sixteen queues of four."""
    ci = np.ascontiguousarray(cam_idx, dtype=np.uint32); pi = np.ascontiguousarray(pt_idx, dtype=np.uint32)
    counts = np.zeros(4, dtype=np.int64)
    L = load()
    nd = 8 if dc == 9 else 17
    rc = L.apexgpu_debug_pair_lists_queued_dc(n_cam, n_pt, len(ci), dc, ptr(ci), ptr(pi), ptr(counts), None, None, None, None, None, None)
    if rc != 0:
        raise LinAlgError(rc, "apexgpu_debug_pair_lists_queued_dc failed")
    recs = np.zeros((counts[0], 4), dtype=np.uint32); chunks = np.zeros((counts[1], 2), dtype=np.int32)
    blocks = np.zeros((counts[2], 4), dtype=np.int64); tasks = np.zeros((counts[3], 2), dtype=np.int32)
    o_index = np.zeros(len(ci), dtype=np.int32); qdesc = np.zeros((counts[1] * nd, 3), dtype=np.int64)
    rc = L.apexgpu_debug_pair_lists_queued_dc(n_cam, n_pt, len(ci), dc, ptr(ci), ptr(pi), ptr(counts), ptr(recs), ptr(chunks), ptr(blocks),
                                              ptr(tasks), ptr(o_index), ptr(qdesc))
    if rc != 0:
        raise LinAlgError(rc, "apexgpu_debug_pair_lists_queued_dc failed")
    return dict(recs=recs, chunks=chunks, blocks=blocks, tasks=tasks, o_index=o_index, qdesc=qdesc.reshape(-1, nd, 3), dc=dc)


def invert_blocks_on_device(blocks: np.ndarray, device: int = 0):
    """The device's eigenvalue-gated 3x3 inverse (row A9) on caller-supplied blocks [n,3,3] -> (inverses, ok)."""
    b = np.ascontiguousarray(blocks, dtype=np.float64).reshape(-1, 9)
    out = np.empty_like(b)
    ok = np.zeros(len(b), dtype=np.int32)
    rc = load().apexgpu_debug_invert_blocks(int(device), len(b), ptr(b), ptr(out), ptr(ok))
    if rc != 0:
        raise LinAlgError(rc, "apexgpu_debug_invert_blocks failed")
    return out.reshape(-1, 3, 3), ok.astype(bool)


def shard_range(pt_idx: np.ndarray, n_pt: int, rank: int, world: int) -> tuple[int, int]:
    """Landmark range owned by `rank` (host arithmetic in the library, no GPU needed)."""
    L = load()
    a = np.ascontiguousarray(pt_idx, dtype=np.uint32)
    lo, hi = C.c_int64(), C.c_int64()
    rc = L.apexgpu_shard_range(n_pt, a.shape[0], a.ctypes.data_as(C.c_void_p), rank, world, C.byref(lo), C.byref(hi))
    if rc != 0:
        raise LinAlgError(rc, "apexgpu_shard_range")
    return lo.value, hi.value


def ptr(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


class Handle:
    """RAII wrapper of an apexgpu_solver*."""

    def __init__(self, n_cam: int, n_pt: int, n_obs: int, mode: int, device: int = 0):
        self.L = load()
        self.h = C.c_void_p()
        rc = self.L.apexgpu_create(n_cam, n_pt, n_obs, mode, device, C.byref(self.h))
        if rc != 0:
            raise LinAlgError(rc, "apexgpu_create failed (no MI355X visible to HIP?)")
        self.n_cam, self.n_pt, self.n_obs, self.mode = n_cam, n_pt, n_obs, mode

    def check(self, rc: int):
        if rc != 0:
            raise LinAlgError(rc, self.L.apexgpu_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.apexgpu_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PgHandle:
    """RAII wrapper of an apexgpu_pg_solver*."""

    def __init__(self, n_vertices: int, n_edges: int, device: int = 0):
        self.L = load()
        self.h = C.c_void_p()
        rc = self.L.apexgpu_pg_create(n_vertices, n_edges, device, C.byref(self.h))
        if rc != 0:
            raise LinAlgError(rc, "apexgpu_pg_create failed (no MI355X visible to HIP?)")
        self.n_vertices, self.n_edges = n_vertices, n_edges

    def check(self, rc: int):
        if rc != 0:
            raise LinAlgError(rc, self.L.apexgpu_pg_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.apexgpu_pg_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
