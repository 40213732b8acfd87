"""The distributed Cholesky of the reduced camera system (csrc/tile_plan.h: subtrees of the nested-dissection
elimination tree on their owner ranks, shared top, two vector exchanges in the triangular sweeps) driven in LOCKSTEP:
the `world` ranks of one sharded problem live in this process on one GPU and the library plays the all-reduces between
the phases (apexgpu_debug_lockstep_solve).  The production path runs the same phases with ncclAllReduce on the same
buffers.  Every result is held to the single-rank solve of the same system."""
import os

import numpy as np
import pytest

import apex_solver_amd as pkg
from apex_solver_amd.capi import LinAlgError
from apex_solver_amd.solver import GpuSchurComplementSolver, OptimizationType, Problem

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.linalg.norm(np.ravel(a) - np.ravel(b)) / max(np.linalg.norm(np.ravel(b)), 1e-300))


def make(d, mode, shard=None, opts=()):
    ot = OptimizationType.SelfCalibration if mode == "selfcal" else OptimizationType.BundleAdjustment
    prob = Problem.bundle_adjustment(d, ot, 1.0)
    s = GpuSchurComplementSolver(0)
    for k, v in opts:
        s.with_option(k, v)
    if shard:
        s.with_shard(*shard)
    s.initialize_structure(prob)
    s.set_parameters(d.poses, d.intr, d.points)
    return prob, s


def single_rank_reference(d, mode, lam, scaling=None):
    prob, s = make(d, mode)
    if scaling is not None:
        s.apply_column_scaling(scaling)
    step = s.solve_augmented_equation(lam)
    _, gred = s.get_schur(want_S=False)
    return prob, s, step, gred


def lockstep(d, mode, world, lam, scaling=None, opts=()):
    ranks = [make(d, mode, shard=(r, world), opts=opts)[1] for r in range(world)]
    if scaling is not None:
        for s in ranks:
            s.apply_column_scaling(scaling)
    GpuSchurComplementSolver.lockstep_solve(ranks, lam)
    return ranks


@pytest.mark.parametrize("tree", [1, 0], ids=["tree-sharded", "range-sharded"])
@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_lockstep_distributed_solve_matches_single_rank(world, tree):
    """tree-sharded (default): a landmark belongs to the rank whose columns it touches -- no reduction of S at all;
    range-sharded: contiguous landmark ranges, every column's tiles reduced to the owner."""
    d = pkg.synthetic.make_problem(1500, 30000, 3, 7, config_id=310)     # 94 tile rows, banded covisibility
    lam = 1e-3
    prob, s1, step1, gred = single_rank_reference(d, "selfcal", lam)
    ranks = lockstep(d, "selfcal", world, lam, opts=(("tree_sharding", tree),))
    infos = [s.info() for s in ranks]
    assert all(i["tree_sharded"] == bool(tree) for i in infos)
    print("observations per rank", [i["local_obs"] for i in infos])
    print("world", world, "tile rows", infos[0]["tile_rows"], "top columns", infos[0]["dist_top_columns"],
          "local fractions", [round(i["dist_local_fraction"], 3) for i in infos])
    assert infos[0]["dist_top_columns"] > 0 and all(i["dist_top_columns"] == infos[0]["dist_top_columns"] for i in infos)
    assert abs(sum(i["dist_local_fraction"] for i in infos) - 1.0) < 1e-9
    # the cut balances as far as a 94-column tree allows (the estimate trades balance against the replicated top)
    assert max(i["dist_local_fraction"] for i in infos) < {2: 0.6, 3: 0.55, 4: 0.35, 8: 0.3}[world]
    nc = prob.layout.cam_dof
    lay = prob.layout
    steps = [s.export_step()[0] for s in ranks]
    for r in range(1, world):                                              # every rank ends with the same camera step
        assert np.array_equal(steps[r][:nc], steps[0][:nc])
    # camera part: backward error against the single-rank S (explicit tiles), and agreement with its step
    Sx, _ = s1.schur_matvec(lam, steps[0][:nc], implicit=False)
    S1, _ = s1.schur_matvec(lam, step1[:nc], implicit=False)
    r_dist = np.linalg.norm(Sx - gred) / np.linalg.norm(gred)
    r_one = np.linalg.norm(S1 - gred) / np.linalg.norm(gred)
    print("residual distributed / single", r_dist, r_one, "step difference", rel(steps[0][:nc], step1[:nc]))
    assert r_dist < 10 * max(r_one, 1e-13)
    assert rel(steps[0][:nc], step1[:nc]) < 1e-7
    # landmark part: each rank back-substitutes its own range
    owned = np.zeros(d.n_pt, dtype=int)
    for r, s in enumerate(ranks):
        m = s.owned_landmarks()
        if not tree:
            lo, hi = pkg.capi.shard_range(d.pt_idx, d.n_pt, r, world)
            assert np.array_equal(np.flatnonzero(m), np.arange(lo, hi))
        owned += m
        cols = (lay.pt_col[m][:, None] + np.arange(3)[None]).ravel()
        assert rel(steps[r][cols], step1[cols]) < 1e-6
    assert np.all(owned == 1)
    for s in ranks + [s1]:
        s.close()


def test_lockstep_with_jacobi_scaling_and_ba_mode():
    d = pkg.synthetic.make_problem(1200, 20000, 3, 7, config_id=311)
    lam = 1e-2
    prob, s0 = make(d, "ba")
    scal = 1.0 / (1.0 + s0.compute_column_norms())
    s0.close()
    prob, s1, y1, gred = single_rank_reference(d, "ba", lam, scaling=scal)
    ranks = lockstep(d, "ba", 2, lam, scaling=scal)
    nc = prob.layout.cam_dof
    y = ranks[0].export_step()[0]
    assert rel(y[:nc], y1[:nc]) < 1e-7
    for s in ranks + [s1]:
        s.close()


def test_lockstep_needs_a_distributed_plan():
    d = pkg.synthetic.make_problem(1200, 20000, 3, 7, config_id=312)
    ranks = [make(d, "selfcal", shard=(r, 2), opts=(("dist_factor", 0),))[1] for r in range(2)]
    assert ranks[0].info()["dist_top_columns"] == 0
    with pytest.raises(LinAlgError) as e:
        GpuSchurComplementSolver.lockstep_solve(ranks, 1e-3)
    assert e.value.kind == "InvalidState"
    for s in ranks:
        s.close()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_distributed_schedule_on_one_rank_equals_the_plain_schedule(world):
    """Option dist_selftest: the tree is cut as for `world` ranks, this rank owns every subtree, the exchanges are
    no-ops -- the two-phase factorisation and the phased sweeps (graphs 0, 3, 4, 5) run through the NORMAL solve and LM
    entry points and must reproduce the plain level schedule."""
    from apex_solver_amd.solver import LevenbergMarquardt, LevenbergMarquardtConfig
    d = pkg.synthetic.make_problem(1500, 30000, 3, 7, config_id=313)
    lam = 1e-4
    prob, s1, step1, gred = single_rank_reference(d, "selfcal", lam)
    _, s = make(d, "selfcal", opts=(("dist_selftest", world),))
    info = s.info()
    assert info["dist_top_columns"] > 0 and not info["tree_sharded"]
    step = s.solve_augmented_equation(lam)
    nc = prob.layout.cam_dof
    Sx, _ = s1.schur_matvec(lam, step[:nc], implicit=False)
    assert np.linalg.norm(Sx - gred) / np.linalg.norm(gred) < 1e-13
    assert rel(step, step1) < 1e-7
    # second solve on the same handle replays the captured graphs
    assert rel(s.solve_augmented_equation(lam), step) < 1e-9
    for x in (s, s1):
        x.close()


@pytest.mark.parametrize("world", [2, 3])
def test_lockstep_on_a_forest_with_a_partial_last_tile(world):
    """Two disconnected captures (the elimination tree is a forest) and 9 n_cam not a multiple of 144: the last tile column
    is a subtree root below the shared top, owned by some rank q; the identity on its padding rows must sit on rank q's
    copy -- the only one that is ever factorised (with the identity on rank 0 only, rank q met zero pivots)."""
    a = pkg.synthetic.make_problem(1408, 24000, 3, 7, config_id=321)           # 88 whole tiles
    b = pkg.synthetic.make_problem(41, 800, 3, 7, config_id=322, window=16)    # a second, small capture: tiles 88 .. 90
    d = pkg.synthetic.BAProblemData(
        poses=np.concatenate([a.poses, b.poses]), intr=np.concatenate([a.intr, b.intr]), points=np.concatenate([a.points, b.points]),
        cam_idx=np.concatenate([a.cam_idx, b.cam_idx + a.n_cam]), pt_idx=np.concatenate([a.pt_idx, b.pt_idx + a.n_pt]),
        obs_uv=np.concatenate([a.obs_uv, b.obs_uv]), truth_poses=np.concatenate([a.truth_poses, b.truth_poses]),
        truth_intr=np.concatenate([a.truth_intr, b.truth_intr]), truth_points=np.concatenate([a.truth_points, b.truth_points]),
        name="two-captures")
    assert (9 * d.n_cam) % 144 != 0
    hs = pkg.capi.host_structure(d.n_cam, d.n_pt, d.cam_idx, d.pt_idx, rank=0, world=world)
    assert hs["tile_owner"][-1] > 0, "the last tile column must belong to a rank other than 0 for this test to bite"
    lam = 1e-3
    prob, s1, step1, gred = single_rank_reference(d, "selfcal", lam)
    ranks = lockstep(d, "selfcal", world, lam)
    infos = [s.info() for s in ranks]
    print("world", world, "top columns", infos[0]["dist_top_columns"], "local fractions", [round(i["dist_local_fraction"], 3) for i in infos],
          "tree sharded", [i["tree_sharded"] for i in infos])
    nc = prob.layout.cam_dof
    steps = [s.export_step()[0] for s in ranks]
    for r in range(1, world):
        assert np.array_equal(steps[r][:nc], steps[0][:nc])
    Sx, _ = s1.schur_matvec(lam, steps[0][:nc], implicit=False)
    r_dist = np.linalg.norm(Sx - gred) / np.linalg.norm(gred)
    print("residual", r_dist, "step difference", rel(steps[0][:nc], step1[:nc]))
    assert r_dist < 1e-12 and rel(steps[0][:nc], step1[:nc]) < 1e-7
    for s in ranks + [s1]:
        s.close()


def test_factorisation_schedule_switches_agree():
    """The streams and events of TilePlan::enqueue_factor (U1d / U1o / U2a / U2b1 / U2b2 on four streams, flood gates) only
    reorder launches that do not touch the same tiles; every target tile sees its updates in the same order whatever the
    switches say.  A missing edge would show as a wrong or irreproducible step: all schedules against the fully serial one
    (everything on the main stream), twice each, on a banded problem with ~90 tile rows."""
    import apex_solver_amd as pkg
    from apex_solver_amd.solver import GpuSchurComplementSolver, OptimizationType, Problem

    d, _, _ = pkg.datasets.load_named("final-13682", 0.1)
    prob = Problem.bundle_adjustment(d, OptimizationType.SelfCalibration, 1.0)

    def step(opts):
        s = GpuSchurComplementSolver(0)
        for k, v in opts.items():
            s.with_option(k, v)
        s.initialize_structure(prob)
        s.set_parameters(d.poses, d.intr, d.points)
        out = [s.solve_augmented_equation(1e-3).copy(), s.solve_augmented_equation(1e-3).copy()]
        info = s.info()
        s.close()
        return out, info

    (ref, _), info = step({"update_overlap": 0, "flood_gate": 0, "two_side": 0})
    assert info["etree_levels"] >= 8
    nrm = np.linalg.norm(ref)
    for opts in ({"two_side": 2}, {"two_side": 2, "flood_gate": 2}, {"two_side": 0, "flood_gate": 2}, {"two_side": 2, "flood_gate": 0},
                 {"two_side": 2, "split_u1": 0}, {"update_overlap": 1, "two_side": 0},
                 # round 4: the top of the tree by level launches only / as one dataflow launch down to wide groups (the default
                 # lets a cost model choose)
                 {"factor_flow": 0}, {"factor_flow": 64}, {"factor_flow": 64, "factor_flow_rows": 1000},
                 # one host wait per solve / three; the step statistics and the trial point behind the solve / on request
                 {"one_wait": 0}, {"one_wait": 0, "factor_flow": 0}, {"eager_step_eval": 0}, {"graphs": 0}):
        (a, b), _ = step(opts)
        # Bit for bit (round 4): with the queued pair layout S is assembled without atomics on this shape (no block longer than
        # a piece, no camera that sees a landmark twice), every schedule adds a tile's updates in the same order, the dataflow
        # launch and the triangular panel skip change no finite value, and the dataflow sweeps fold in list order.
        assert np.array_equal(a, ref) and np.array_equal(b, a), (opts, np.linalg.norm(a - ref) / nrm)
