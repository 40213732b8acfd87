"""The triangular sweeps as one dataflow launch each (k_tri_fwd_flow / k_tri_bwd_flow, csrc/chol_kernels.hip): same step as
the level-by-level sweeps, bitwise reproducible (no atomics on data, fixed fold order), on a deep elimination tree
(many levels, rows with dozens of products) and on the pose-graph path."""
import numpy as np
import pytest

import apex_solver_amd as pkg
from apex_solver_amd.pose_graph import GpuSparseCholeskySolver, PoseGraphProblem
from apex_solver_amd.solver import GpuSchurComplementSolver, OptimizationType, Problem

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


@pytest.mark.parametrize("nd", [1, 0])
def test_ba_dataflow_sweeps_match_level_sweeps_and_are_reproducible(nd):
    d = pkg.synthetic.make_problem(1200, 40000, 3, 8, config_id=83)
    steps = {}
    for flow in (1, 0):
        prob = Problem.bundle_adjustment(d, OptimizationType.SelfCalibration, 1.0)
        s = GpuSchurComplementSolver(0).with_option("nested_dissection", nd).with_option("tri_dataflow", flow).initialize_structure(prob)
        s.set_parameters(d.poses, d.intr, d.points)
        a = s.solve_augmented_equation(1e-3).copy()
        b = s.solve_augmented_equation(1e-3).copy()
        info = s.info()
        steps[flow] = (a, b)
        s.close()
    print(nd, info, rel(steps[1][0], steps[0][0]))
    assert np.array_equal(steps[1][0], steps[1][1]), "dataflow sweeps must be bitwise reproducible"
    # two correct triangular solves of the same factor differ by rounding amplified by cond(S): see test_gpu_parity
    assert rel(steps[1][0], steps[0][0]) < 1e-9
    assert np.all(np.isfinite(steps[1][0]))


def test_pose_graph_dataflow_sweeps_match_level_sweeps():
    d = pkg.synthetic.make_sphere(30, 40)
    steps = {}
    for flow in (1, 0):
        s = GpuSparseCholeskySolver(0).with_option("tri_dataflow", flow).initialize_structure(PoseGraphProblem.pose_graph(d))
        s.set_parameters(d.poses)
        a = s.solve_augmented_equation(1e-3).copy()
        b = s.solve_augmented_equation(1e-3).copy()
        steps[flow] = (a, b)
        s.close()
    # (H is assembled with fp64 atomics here: two solves differ by last bits times cond(H) whatever the sweeps do)
    assert rel(steps[1][0], steps[1][1]) < 1e-9
    assert rel(steps[1][0], steps[0][0]) < 1e-9


# ---- a sweep that gives up fails (and is repaired inside) the solve it belongs to ------------------------------------------
@pytest.mark.parametrize("which", [1, 2], ids=["forward", "backward"])
def test_sweep_timeout_is_caught_in_the_same_solve_and_repaired(which):
    """"debug_poison_sweep": one block's counter is made unreachable, so the dataflow sweep runs into its spin limit, raises
    the error word and leaves a wrong x.  The SAME solve_augmented must notice (error word posted to the host behind the
    sweeps), repeat the triangular solve level by level and return the right step; the event is counted and the handle
    stays on the level sweeps."""
    d = pkg.synthetic.make_problem(400, 12000, 3, 8, config_id=84)
    prob = Problem.bundle_adjustment(d, OptimizationType.SelfCalibration, 1.0)
    ref = GpuSchurComplementSolver(0).with_option("tri_dataflow", 0).initialize_structure(prob)
    ref.set_parameters(d.poses, d.intr, d.points)
    want = ref.solve_augmented_equation(1e-3).copy()
    ref.close()
    s = GpuSchurComplementSolver(0).initialize_structure(prob)
    s.set_parameters(d.poses, d.intr, d.points)
    good = s.solve_augmented_equation(1e-3).copy()
    c0 = s.counters()
    assert (c0["sweep_timeouts"], c0["tri_dataflow"], c0["factor_flow_timeouts"]) == (0, True, 0)
    s.set_option("debug_poison_sweep", which)
    got = s.solve_augmented_equation(1e-3).copy()          # ~2 s: the poisoned wait runs to its limit
    c = s.counters()
    print("sweep", which, "counters after the poisoned solve:", c, "vs level sweeps", rel(got, want), "vs dataflow", rel(got, good))
    assert (c["sweep_timeouts"], c["tri_dataflow"], c["factor_flow_timeouts"]) == (1, False, 0)
    # the repaired solve is a level-sweep solve of the same factor (its forward sweep adds into shared ancestor blocks
    # with atomics, so two of them agree to rounding times cond(S), not bit for bit); the poisoned sweep's x was garbage
    assert rel(got, want) < 1e-9 and rel(got, good) < 1e-9
    # the step, the statistics and the trial cost of that call are the repaired ones
    gn, sn, pred = s.step_stats()
    assert np.isfinite([gn, sn, pred]).all() and pred > 0
    again = s.solve_augmented_equation(1e-3).copy()         # the handle stays on the level sweeps
    assert rel(again, want) < 1e-9 and s.counters()["sweep_timeouts"] == 1
    s.close()


def test_pose_graph_sweep_timeout_is_repaired():
    d = pkg.synthetic.make_sphere(30, 40)
    s = GpuSparseCholeskySolver(0).initialize_structure(PoseGraphProblem.pose_graph(d))
    s.set_parameters(d.poses)
    good = s.solve_augmented_equation(1e-3).copy()
    s.set_option("debug_poison_sweep", 1)
    got = s.solve_augmented_equation(1e-3).copy()
    c = s.counters()
    assert (c["sweep_timeouts"], c["tri_dataflow"], c["factor_flow_timeouts"]) == (1, False, 0)
    assert rel(got, good) < 1e-9 and np.all(np.isfinite(got))
    s.close()


@pytest.mark.parametrize("busy", [224, 248])
def test_dataflow_sweeps_make_progress_while_most_cus_are_blocked(busy):
    """Forward progress under contention: `busy` of the 256 compute units are taken by workgroups of another stream that hold
    all of their LDS for 40 ms ("debug_occupy_cus").  The sweeps' one-workgroup-per-tile launches then run on the remaining
    CUs only -- far fewer resident workgroups than tasks, every wait must still be for an EARLIER workgroup.  No time-out,
    same bits as on the idle chip."""
    d = pkg.synthetic.make_problem(1200, 40000, 3, 8, config_id=83)
    prob = Problem.bundle_adjustment(d, OptimizationType.SelfCalibration, 1.0)
    s = GpuSchurComplementSolver(0).initialize_structure(prob)
    s.set_parameters(d.poses, d.intr, d.points)
    idle = s.solve_augmented_equation(1e-3).copy()
    for rep in range(3):
        s.set_option("debug_occupy_cus", busy)              # returns once the blocking workgroups are resident
        got = s.solve_augmented_equation(1e-3).copy()
        c0 = s.counters()
        assert (c0["sweep_timeouts"], c0["tri_dataflow"], c0["factor_flow_timeouts"]) == (0, True, 0)
        # (S is assembled deterministically on this shape and the sweeps fold in list order: bitwise equal)
        assert np.array_equal(got, idle)
    s.close()


# ---- the dataflow launch of the factorisation's top groups: a launch that gives up is repaired inside the same solve ---------
def test_factor_flow_timeout_is_caught_in_the_same_solve_and_repaired():
    """"debug_poison_factor": one version counter of k_factor_flow is made unreachable, the units downstream run into the spin
    limit (one time-out, the rest leave at once through the error word) and the tiles are left half updated.  The SAME
    solve_augmented must notice (flag word read with the pivot flag), assemble S again, factorise it with the level launches
    and return the right step; the event is counted and the handle stays on the level launches."""
    d = pkg.datasets.load_named("ladybug-1723", 0.25)[0]
    prob = Problem.bundle_adjustment(d, OptimizationType.SelfCalibration, 1.0)
    ref = GpuSchurComplementSolver(0).with_option("factor_flow", 0).initialize_structure(prob)
    ref.set_parameters(d.poses, d.intr, d.points)
    want = ref.solve_augmented_equation(1e-3).copy()
    assert ref.counters()["factor_flow_groups"] == 0
    ref.close()
    s = GpuSchurComplementSolver(0).with_option("factor_flow", 8).initialize_structure(prob)
    s.set_parameters(d.poses, d.intr, d.points)
    good = s.solve_augmented_equation(1e-3).copy()
    c0 = s.counters()
    assert c0["factor_flow_groups"] >= 2 and c0["factor_flow_timeouts"] == 0, c0
    # (S is assembled with atomics on a few shared blocks: two factorisations agree to rounding times cond(S))
    assert rel(good, want) < 1e-7
    s.set_option("debug_poison_factor", 1)
    got = s.solve_augmented_equation(1e-3).copy()          # a few seconds: the poisoned wait runs to its limit
    c = s.counters()
    print("counters after the poisoned factorisation:", c, "vs level launches", rel(got, want))
    assert c["factor_flow_timeouts"] == 1 and np.all(np.isfinite(got)) and rel(got, want) < 1e-7
    again = s.solve_augmented_equation(1e-3).copy()         # the handle stays on the level launches
    assert rel(again, want) < 1e-7 and s.counters()["factor_flow_timeouts"] == 1
    s.close()
