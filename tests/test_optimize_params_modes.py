"""The other OptimizeParams<POSE, LANDMARK, INTRINSIC> configurations (src/factors/mod.rs:82-101: OnlyPose, OnlyLandmarks,
OnlyIntrinsics, PoseAndIntrinsics, LandmarksAndIntrinsics) -- SURVEY §8(f)4.

Semantics restated (projection_factor.rs:184-296, 306-364): the factor's Jacobian has columns for the optimised blocks
only; the other blocks are constants of the factor.  With the bin's variable set (every pose_*, intr_*, pt_* exists) the
blocks that are not optimised are zero columns of the global Jacobian, and the damped system gives them a zero step.

CPU part: the oracle's mode extension against an INDEPENDENT numpy / scipy restatement (the reduced problem over the
optimised variables only, solved directly).  GPU part: the device against both."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import apex_solver_amd as pkg
import np_ref
from apex_solver_amd.solver import GpuSchurComplementSolver, OptimizationType, Problem

MODES = {"only_pose": OptimizationType.OnlyPose, "only_landmarks": OptimizationType.OnlyLandmarks,
         "only_intrinsics": OptimizationType.OnlyIntrinsics, "pose_and_intrinsics": OptimizationType.PoseAndIntrinsics,
         "landmarks_and_intrinsics": OptimizationType.LandmarksAndIntrinsics}


def rel(a, b):
    a = np.ravel(a); b = np.ravel(b)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def reduced_direct_step(oracle, d, lay, flags, lam):
    """Independent restatement: J of the FULL self-calibration linearisation (all blocks), columns of the optimised blocks
    only, direct sparse solve of (J^T J + lambda I) dx = -J^T r over those variables; zero step elsewhere."""
    full = oracle.from_data(d, lay, mode="selfcal", huber_delta=1.0)
    _, r, Jp, Jl, Ji = full.linearize()
    P, L, I = flags
    J = np_ref.sparse_jacobian(Jp * P, Jl * L, Ji * I, d.cam_idx, d.pt_idx, lay, selfcal=True).tocsc()
    keep = np.zeros(lay.total_dof, dtype=bool)
    for c in range(d.n_cam):
        if P: keep[lay.pose_col[c]:lay.pose_col[c] + 6] = True
        if I: keep[lay.intr_col[c]:lay.intr_col[c] + 3] = True
    if L:
        for l in range(d.n_pt):
            keep[lay.pt_col[l]:lay.pt_col[l] + 3] = True
    Jk = J[:, keep]
    A = (Jk.T @ Jk + lam * sp.identity(Jk.shape[1])).tocsc()
    g = J.T @ r
    step = np.zeros(lay.total_dof)
    step[keep] = spla.spsolve(A, -g[keep])
    return step, g


@pytest.mark.parametrize("mode", sorted(MODES))
def test_oracle_modes_match_the_reduced_problem(oracle, mode):
    d = pkg.synthetic.make_problem(16, 500, 3, 7, config_id=520)
    lay = pkg.layout.reference_column_layout(d.n_cam, d.n_pt)
    lam = 1e-2
    o = oracle.from_data(d, lay, mode=mode, huber_delta=1.0)
    o.linearize()
    ostep, ograd = o.solve_augmented(lam, 0)
    ref, g = reduced_direct_step(oracle, d, lay, MODES[mode].flags, lam)
    assert rel(ograd, g) < 1e-13
    assert rel(ostep, ref) < 1e-9
    P, L, I = MODES[mode].flags
    if not L:
        assert np.abs(ostep[9 * d.n_cam:]).max() == 0.0
    if not P:
        assert all(np.abs(ostep[lay.pose_col[c]:lay.pose_col[c] + 6]).max() == 0.0 for c in range(d.n_cam))
    if not I:
        assert all(np.abs(ostep[lay.intr_col[c]:lay.intr_col[c] + 3]).max() == 0.0 for c in range(d.n_cam))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", sorted(MODES))
def test_gpu_modes_one_iteration_and_lm(oracle, mode):
    d = pkg.synthetic.make_problem(40, 2000, 3, 7, config_id=521)
    ot = MODES[mode]
    prob = Problem.bundle_adjustment(d, ot, 1.0)
    s = GpuSchurComplementSolver(0).initialize_structure(prob)
    s.set_parameters(d.poses, d.intr, d.points)
    lay = prob.layout
    lam = 1e-2
    o = oracle.from_data(d, lay, mode=mode, huber_delta=1.0)
    assert s.compute_cost() == pytest.approx(o.residuals()[0], rel=1e-13)
    o.linearize()
    ostep, ograd = o.solve_augmented(lam, 0)
    ref, g = reduced_direct_step(oracle, d, lay, ot.flags, lam)
    step = s.solve_augmented_equation(lam)
    errs = dict(grad=rel(s.get_gradient(), ograd), step_vs_oracle=rel(step, ostep), step_vs_direct=rel(step, ref))
    print(mode, {k: f"{v:.1e}" for k, v in errs.items()})
    assert errs["grad"] < 1e-12 and errs["step_vs_oracle"] < 1e-8 and errs["step_vs_direct"] < 1e-8
    P, L, I = ot.flags
    if not L:
        assert np.abs(step[9 * d.n_cam:]).max() == 0.0
    o.apply_step(ostep, 1.0)
    assert s.eval_step() == pytest.approx(o.residuals()[0], rel=1e-9)
    s.discard_step()
    # the LM loop on the device against the oracle's loop in the same mode
    from apex_solver_amd.solver import LevenbergMarquardt, LevenbergMarquardtConfig
    res = LevenbergMarquardt.with_config(LevenbergMarquardtConfig().with_max_iterations(6)).optimize(prob)
    o2 = oracle.from_data(d, lay, mode=mode, huber_delta=1.0)
    ores = o2.optimize(oracle.LMConfig.default(max_iterations=6))
    assert res.iterations == ores.iterations and res.status.name == ores.status
    assert np.array_equal(res.history[:, 3], ores.history[:, 3])
    assert np.allclose(res.history[:, 0], ores.history[:, 0], rtol=1e-7)
    assert res.final_cost < res.initial_cost
    s.close()
