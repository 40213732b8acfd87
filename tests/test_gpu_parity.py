"""Parity of the HIP path (through the C ABI) with the CPU oracle and the golden fixtures.
Every test needs a real MI355X: run with `pytest -m gpu`.  Tolerances are written next to
each assertion; integer/index work does not occur on this path, everything is fp64.
"""
import glob
import os

import numpy as np
import pytest

import apex_solver_amd as pkg
import np_ref
import referee
from apex_solver_amd.solver import (GpuSchurComplementSolver, LevenbergMarquardt, LevenbergMarquardtConfig,
                                    OptimizationStatus, OptimizationType, Problem, SchurVariant)

pytestmark = pytest.mark.gpu
# Forward bound on |step - oracle step| / |oracle step| wherever the two are compared.  The north star's 1e-10 is met where
# cond(S) <= ~1e5 (pose graphs, well-damped systems); on gauge-free BA systems (cond 1e9..1e10) two correct Choleskys of
# the same S differ by up to ~1e-9 (SelfCalibration) and ~5e-8 (BundleAdjustment mode: measured 3.4e-8 / 5.3e-8 on
# ba6x40_ba / ragged landmarks at backward errors of 1e-15), so the bound that every case must meet is 1e-7 -- three
# orders tighter than the eps*cond(S) allowance it replaces -- NEXT TO a 1e-13 normwise backward error on the system
# actually solved.
STEP_FORWARD_BOUND = 1e-7
GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ba*.npz")))


def rel(a, b):
    a = np.ravel(a); b = np.ravel(b)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def gpu_solver(d, mode, huber=1.0, variant=SchurVariant.Sparse, fix_first=True, shard=None, options=None):
    ot = OptimizationType.SelfCalibration if mode == "selfcal" else OptimizationType.BundleAdjustment
    prob = Problem.bundle_adjustment(d, ot, huber) if fix_first else Problem(d, ot, huber)
    s = GpuSchurComplementSolver(0).with_variant(variant)
    if shard:
        s.with_shard(*shard)
    for k, v in (options or {}).items():
        s.with_option(k, v)
    s.initialize_structure(prob)
    s.set_parameters(d.poses, d.intr, d.points)
    return prob, s


def oracle_problem(ora, d, prob, mode):
    o = ora.from_data(d, prob.layout, mode=mode, huber_delta=prob.huber_delta if prob.huber_delta else -1.0)
    return o


def jc_to_blocks(jc, dc):
    """GPU camera block [pose(6) | intr(3)] -> (Jpose, Jintr)"""
    jp = jc[:, :, :6]
    ji = jc[:, :, 6:9] if dc == 9 else np.zeros((jc.shape[0], 2, 3))
    return jp, ji


def data_from_golden(g):
    from apex_solver_amd.synthetic import BAProblemData

    return BAProblemData(poses=g["poses0"], intr=g["intr0"], points=g["points0"], cam_idx=g["cam_idx"],
                         pt_idx=g["pt_idx"], obs_uv=g["obs_uv"], name="golden")


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_golden_fixture_iterations(path):
    """Three LM iterations of each committed fixture, stage by stage."""
    g = np.load(path)
    d = data_from_golden(g)
    mode = str(g["mode"])
    prob, s = gpu_solver(d, mode)
    dc = 9 if mode == "selfcal" else 6
    assert s.compute_cost() == pytest.approx(float(g["initial_cost"]), rel=1e-13)
    for it in range(int(g["iters"])):
        lam = float(g[f"it{it}_lambda"])
        # every iteration starts from the fixture's own parameters: the LM trajectory amplifies the
        # eps*cond(S) forward error of the previous step, which is not what is under test here
        s.set_parameters(g[f"it{it}_poses"], g[f"it{it}_intr"], g[f"it{it}_points"])
        assert rel(s.get_residual(), g[f"it{it}_r"]) < 1e-12
        jc, jl = s.get_jacobian_blocks()
        jp, ji = jc_to_blocks(jc, dc)
        assert rel(jp, g[f"it{it}_Jpose"]) < 1e-12 and rel(jl, g[f"it{it}_Jpt"]) < 1e-12
        if dc == 9:
            assert rel(ji, g[f"it{it}_Jintr"]) < 1e-12
        # well-conditioned regime first (lambda = 1e4, cond(S) <= 1e5, fixtures' it*_wc_*): the north star's 1e-10 directly
        wstep = s.solve_augmented_equation(float(g[f"it{it}_wc_lambda"]))
        wS, wgred = s.get_schur()
        werr = dict(S=rel(wS, g[f"it{it}_wc_S"]), gred=rel(wgred, g[f"it{it}_wc_gred"]), step=rel(wstep, g[f"it{it}_wc_step"]),
                    step_exact=rel(wstep, g[f"it{it}_wc_step_exact"]))
        print(os.path.basename(path), "iter", it, "lambda 1e4:", {k: f"{v:.1e}" for k, v in werr.items()})
        assert werr["S"] < 1e-12 and werr["gred"] < 1e-11 and werr["step"] < referee.NORTH_STAR and werr["step_exact"] < referee.NORTH_STAR
        step = s.solve_augmented_equation(lam)
        grad = s.get_gradient()
        S, gred = s.get_schur()
        errs = dict(grad=rel(grad, g[f"it{it}_grad"]), S=rel(S, g[f"it{it}_S"]), gred=rel(gred, g[f"it{it}_gred"]),
                    step=rel(step, g[f"it{it}_step"]))
        # the referee: the device against the EXACT step of the fixture's linearisation, next to the fp64 oracle's error
        e_gpu, e_64 = rel(step, g[f"it{it}_step_exact"]), rel(g[f"it{it}_step"], g[f"it{it}_step_exact"])
        print(os.path.basename(path), "iter", it, f"referee: |gpu - exact| {e_gpu:.2e}  |fp64 oracle - exact| {e_64:.2e}")
        referee.RECORD.append((f"{os.path.basename(path)[:-4]} it{it}", e_gpu, e_64))
        assert e_gpu <= referee.FP64_ENVELOPE, (e_gpu, e_64)
        print(os.path.basename(path), "iter", it, {k: f"{v:.1e}" for k, v in errs.items()})
        assert errs["grad"] < 1e-12 and errs["S"] < 1e-12 and errs["gred"] < 1e-11
        # The step is held to (i) a normwise backward error of 1e-13 on S dc = g_red and (ii) a FIXED forward bound of
        # 1e-8 against the fixture (two correct fp64 Choleskys of the same S differ by ~eps*cond(S); cond is 1e9..1e10
        # here because the gauge is only damped -- the measured value is printed above and recorded in DESIGN.md §2)
        Sg = g[f"it{it}_S"]
        assert errs["step"] < STEP_FORWARD_BOUND, errs
        nc = prob.layout.cam_dof
        bwd = np.linalg.norm(Sg @ step[:nc] - g[f"it{it}_gred"]) / (referee.sym_norm2(Sg) * np.linalg.norm(step[:nc]) + np.linalg.norm(g[f"it{it}_gred"]))
        assert bwd < 1e-13, bwd
        gn, sn, pred = s.step_stats()
        assert gn == pytest.approx(np.linalg.norm(g[f"it{it}_grad"]), rel=1e-12)
        assert pred == pytest.approx(float(g[f"it{it}_pred"]), rel=1e-7)
        trial = s.eval_step()
        # the trial point moves with the step's forward error (<= ~1e-8 relative here)
        assert trial == pytest.approx(float(g[f"it{it}_new_cost"]), rel=1e-6)
        if bool(g[f"it{it}_accepted"]):
            s.commit_step()
        else:
            s.discard_step()
    poses, intr, pts = s.get_parameters()
    assert rel(poses, g["poses_end"]) < 1e-8 and rel(pts, g["points_end"]) < 1e-7 and rel(intr, g["intr_end"]) < 1e-8
    s.close()


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_golden_lm_history(path):
    g = np.load(path)
    d = data_from_golden(g)
    mode = str(g["mode"])
    ot = OptimizationType.SelfCalibration if mode == "selfcal" else OptimizationType.BundleAdjustment
    prob = Problem.bundle_adjustment(d, ot, 1.0)
    res = LevenbergMarquardt.with_config(LevenbergMarquardtConfig().with_max_iterations(8)).optimize(prob)
    hg = g["lm_history"]
    assert res.status.name == str(g["lm_status"]) and res.iterations == int(g["lm_iterations"])
    assert np.array_equal(res.history[:, 3], hg[:, 3])            # accept / reject pattern
    assert np.allclose(res.history[:, 0], hg[:, 0], rtol=1e-7)    # cost per iteration
    assert np.allclose(res.history[:, 1], hg[:, 1], rtol=1e-4)    # damping (depends on rho)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["selfcal", "ba"])
@pytest.mark.parametrize("shape", [(40, 2000, 3, 7), (130, 4000, 3, 9)], ids=["40x2000", "130x4000"])
def test_one_iteration_vs_oracle(oracle, mode, shape):
    """Seeded synthetic problems (several S tiles, symbolic fill) against the oracle."""
    n_cam, n_pt, klo, khi = shape
    d = pkg.synthetic.make_problem(n_cam, n_pt, klo, khi, config_id=200 + n_cam)
    prob, s = gpu_solver(d, mode)
    o = oracle_problem(oracle, d, prob, mode)
    assert s.compute_cost() == pytest.approx(o.residuals()[0], rel=1e-13)
    lam = 1e-3
    o.linearize()
    ostep, ograd, oS, ogred = o.solve_augmented(lam, 0, want_schur=True)
    step = s.solve_augmented_equation(lam)
    S, gred = s.get_schur()
    hinv, gl = s.get_landmark_blocks()
    errs = dict(grad=rel(s.get_gradient(), ograd), S=rel(S, oS), gred=rel(gred, ogred), step=rel(step, ostep))
    print(mode, shape, {k: f"{v:.1e}" for k, v in errs.items()}, s.info())
    assert errs["grad"] < 1e-12 and errs["S"] < 1e-12 and errs["gred"] < 1e-10
    # S dc = g_red holds to working precision whatever the conditioning ...
    nc = prob.layout.cam_dof
    bwd = np.linalg.norm(oS @ step[:nc] - ogred) / (referee.sym_norm2(oS) * np.linalg.norm(step[:nc]) + np.linalg.norm(ogred))
    assert bwd < 1e-13
    # ... and the step agrees with the oracle's within the fixed forward bound (north star: 1e-10 where cond(S) allows)
    print("step vs oracle", errs["step"], "cond(S)", np.linalg.cond(oS), "meets 1e-10:", errs["step"] < 1e-10)
    assert errs["step"] < STEP_FORWARD_BOUND, errs["step"]
    # ... and, the actual pass/fail line: as close to the EXACT step as the fp64 oracle is (tests/referee.py)
    dc = 9 if mode == "selfcal" else 6
    referee.check_step(o, s, step, ostep, lam, dc, label=f"{mode} {shape}")
    # trial point
    o.apply_step(ostep, 1.0)
    assert s.eval_step() == pytest.approx(o.residuals()[0], rel=1e-9)
    s.discard_step()
    o.apply_step(ostep, -1.0); o.linearize()
    # well-conditioned regime: 1e-10 against the fp64 oracle outright
    referee.check_well_conditioned(o, s, dc, label=f"{mode} {shape}")
    s.close()


@pytest.mark.parametrize("mode", ["selfcal", "ba"])
def test_pcg_variant_vs_oracle(oracle, mode):
    d = pkg.synthetic.make_problem(30, 1500, 3, 7, config_id=77)
    prob, s = gpu_solver(d, mode, variant=SchurVariant.Iterative)
    o = oracle_problem(oracle, d, prob, mode)
    o.linearize()
    ostep, ograd, oS, ogred = o.solve_augmented(1e-3, 1, want_schur=True)
    step = s.solve_augmented_equation(1e-3)
    nc = prob.layout.cam_dof
    # both stop on |r| < 1e-6*max(|b|,1) (explicit_schur.rs:684) or after 200 iterations
    print("pcg iterations gpu/oracle", s.info()["pcg_iterations"], o.last_pcg_iters)
    assert abs(s.info()["pcg_iterations"] - o.last_pcg_iters) <= 2
    r_gpu = np.linalg.norm(oS @ step[:nc] - ogred); r_ora = np.linalg.norm(oS @ ostep[:nc] - ogred)
    assert r_gpu < 10 * max(r_ora, 1e-6 * max(np.linalg.norm(ogred), 1.0))
    # the SOLUTION, not only the iteration count: both iterations run to 1e-13 (a few hundred to ~1,500 steps on this size) and
    # the two steps must agree -- to 1e-8 at lambda = 1e-3 (cond(S) ~ 1e9; the oracle's own PCG lands 2e-10 from its Cholesky
    # there) and to 1e-10 at lambda = 1e4
    s.with_cg_params(5000, 1e-13); o.set_cg_params(5000, 1e-13)
    for lam, tol in ((1e-3, 1e-8), (1e4, 1e-10)):
        ostep, _ = o.solve_augmented(lam, 1)
        ochol, _ = o.solve_augmented(lam, 0)
        step = s.solve_augmented_equation(lam)
        print(f"lambda {lam:g}: pcg iterations gpu/oracle {s.info()['pcg_iterations']}/{o.last_pcg_iters}  step gpu vs oracle pcg {rel(step, ostep):.2e}"
              f"  vs oracle Cholesky {rel(step, ochol):.2e}")
        assert rel(step, ostep) < tol and rel(step, ochol) < 10 * tol
    s.close()


@pytest.mark.parametrize("mode", ["selfcal", "ba"])
def test_implicit_schur_pcg_vs_oracle(oracle, mode):
    """A18 / BASELINE configs[4] semantics: matrix-free PCG with the Schur-Jacobi preconditioner
    (IterativeSchurSolver, implicit_schur.rs) against the oracle's restatement of the same iteration."""
    d = pkg.synthetic.make_problem(30, 1500, 3, 7, config_id=78)
    prob, s = gpu_solver(d, mode, variant=SchurVariant.Implicit)
    s.with_cg_params(500, 1e-9)
    o = oracle_problem(oracle, d, prob, mode)
    o.set_cg_params(500, 1e-9)
    o.linearize()
    ostep, ograd, oS, ogred = o.solve_augmented(1e-3, 0, want_schur=True)
    istep, _ = o.solve_augmented(1e-3, 2)
    step = s.solve_augmented_equation(1e-3)
    nc = prob.layout.cam_dof
    print("implicit pcg iterations gpu/oracle", s.info()["pcg_iterations"], o.last_pcg_iters)
    assert abs(s.info()["pcg_iterations"] - o.last_pcg_iters) <= max(3, o.last_pcg_iters // 20)
    assert rel(s.get_gradient(), ograd) < 1e-12
    r_gpu = np.linalg.norm(oS @ step[:nc] - ogred); r_ora = np.linalg.norm(oS @ istep[:nc] - ogred)
    assert r_gpu < 10 * max(r_ora, 1e-9 * max(np.linalg.norm(ogred), 1.0))
    # landmark part: back-substitution of the camera step it found
    lay = prob.layout
    A = np.linalg.norm(step - istep) / np.linalg.norm(istep)
    print("step vs oracle implicit", A, "vs Cholesky", np.linalg.norm(step - ostep) / np.linalg.norm(ostep))
    gn, sn, pred = s.step_stats()
    nc_cost = s.eval_step()
    assert nc_cost < s.compute_cost() or pred > 0
    s.discard_step()
    # the SOLUTION of the matrix-free iteration, run to 1e-13 on both sides, where it determines the step: lambda = 1e4
    # (at 1e-3 its stopping rule, relative to max(|b|, 1), leaves 1e-6 in the oracle itself)
    s.with_cg_params(5000, 1e-13); o.set_cg_params(5000, 1e-13)
    istep, _ = o.solve_augmented(1e4, 2)
    ochol, _ = o.solve_augmented(1e4, 0)
    step = s.solve_augmented_equation(1e4)
    print(f"lambda 1e4: implicit pcg iterations gpu/oracle {s.info()['pcg_iterations']}/{o.last_pcg_iters}  step vs oracle implicit {rel(step, istep):.2e}"
          f"  vs oracle Cholesky {rel(step, ochol):.2e}")
    assert rel(step, istep) < 1e-9 and rel(step, ochol) < 1e-9
    s.close()


@pytest.mark.parametrize("mode", ["selfcal", "ba"])
def test_schur_matvec_both_forms_vs_oracle(oracle, mode):
    d = pkg.synthetic.make_problem(30, 1500, 3, 7, config_id=80)
    prob, s = gpu_solver(d, mode)
    o = oracle_problem(oracle, d, prob, mode)
    o.linearize()
    _, _, oS, _ = o.solve_augmented(1e-2, 0, want_schur=True)
    x = np.random.default_rng(0).normal(size=prob.layout.cam_dof)
    ye, yi = s.schur_matvec(1e-2, x)
    ref = oS @ x
    assert rel(ye, ref) < 1e-12 and rel(yi, ref) < 1e-12
    s.close()


def test_full_size_explicit_and_matrix_free_schur_agree():
    """BASELINE configs[3]/[4] headline shape at FULL size (final-13682: 29 M observations, 1.2e8 camera-pair
    blocks): S x through the tiles built by k_cam_reduce + k_schur_pairs_r equals S x through the matrix-free
    operator -- two independent code paths over all observations -- and S is symmetric positive definite on
    the probes (x.Sy == y.Sx, x.Sx > 0)."""
    d = pkg.synthetic.make_named("final-13682")
    prob, s = gpu_solver(d, "selfcal")
    rng = np.random.default_rng(1)
    x = rng.normal(size=prob.layout.cam_dof); y = rng.normal(size=prob.layout.cam_dof)
    sx_e, sx_i = s.schur_matvec(1e-3, x)
    assert rel(sx_e, sx_i) < 1e-11
    sy_e, _ = s.schur_matvec(1e-3, y, implicit=False)
    assert abs(x @ sy_e - y @ sx_e) <= 1e-11 * abs(x @ sy_e)
    assert x @ sx_e > 0 and y @ sy_e > 0
    # and one LM iteration at full size behaves: positive predicted reduction, cost goes down
    c0 = s.compute_cost()
    s.solve_augmented_equation(1e-3, want_step=False)
    gn, sn, pred = s.step_stats()
    c1 = s.eval_step()
    assert pred > 0 and c1 < c0 and np.isfinite(gn) and np.isfinite(sn)
    s.discard_step()
    s.close()


def test_implicit_variant_lm_converges():
    d = pkg.synthetic.make_problem(40, 3000, 3, 7, config_id=79)
    prob = Problem.bundle_adjustment(d, OptimizationType.SelfCalibration, 1.0)
    res = {}
    for v in (SchurVariant.Sparse, SchurVariant.Implicit):
        cfg = LevenbergMarquardtConfig.for_bundle_adjustment().with_schur_variant(v).with_max_iterations(8)
        lm = LevenbergMarquardt.with_config(cfg)
        s = GpuSchurComplementSolver(0).with_cg_params(500, 1e-9)
        res[v] = lm.optimize(prob, solver=s)
        s.close()
    a, b = res[SchurVariant.Sparse], res[SchurVariant.Implicit]
    assert b.final_cost < 0.5 * b.initial_cost
    assert abs(a.final_cost - b.final_cost) <= 1e-3 * a.final_cost


def test_full_normal_equations_from_exported_blocks():
    """Size-independent property: with J rebuilt (scipy.sparse) from the blocks the GPU exports, the
    returned step solves (J^T J + lambda I) dx = -J^T r and the returned gradient is J^T r."""
    d = pkg.synthetic.make_problem(300, 20000, 3, 8, config_id=31)
    for mode in ("selfcal", "ba"):
        prob, s = gpu_solver(d, mode)
        dc = 9 if mode == "selfcal" else 6
        lam = 1e-2
        step = s.solve_augmented_equation(lam)
        jc, jl = s.get_jacobian_blocks()
        jp, ji = jc_to_blocks(jc, dc)
        r = s.get_residual()
        J = np_ref.sparse_jacobian(jp, jl, ji, d.cam_idx, d.pt_idx, prob.layout, selfcal=(mode == "selfcal"))
        g = J.T @ r
        assert rel(s.get_gradient(), g) < 1e-12
        A = (J.T @ J) + lam * np_ref.sp.identity(J.shape[1])
        bwd = np.linalg.norm(A @ step + g) / (np_ref.spla.norm(A) * np.linalg.norm(step) + np.linalg.norm(g))
        print(mode, "normal-equation backward error", bwd, s.info())
        assert bwd < 1e-13
        gn, sn, pred = s.step_stats()
        assert gn == pytest.approx(np.linalg.norm(g), rel=1e-12) and sn == pytest.approx(np.linalg.norm(step), rel=1e-12)
        assert pred == pytest.approx(0.5 * step @ (lam * step - g), rel=1e-10)
        s.close()


# ---- edge cases -----------------------------------------------------------------------------------
def _custom(n_cam, n_pt, cam_lists, seed=5):
    """A problem with explicit per-landmark camera lists (duplicates allowed)."""
    base = pkg.synthetic.make_problem(n_cam, n_pt, 3, 3, config_id=seed)
    cam_idx, pt_idx = [], []
    for l, cams in enumerate(cam_lists):
        cam_idx += list(cams); pt_idx += [l] * len(cams)
    cam_idx = np.asarray(cam_idx, dtype=np.uint32); pt_idx = np.asarray(pt_idx, dtype=np.uint32)
    rng = np.random.default_rng(seed)
    # shuffle the factor order: the caller's order is arbitrary, the library sorts by landmark
    perm = rng.permutation(len(cam_idx))
    cam_idx, pt_idx = cam_idx[perm], pt_idx[perm]
    uv = pkg.synthetic.project_bal(base.truth_poses[cam_idx], base.truth_intr[cam_idx], base.truth_points[pt_idx])
    uv = uv + rng.normal(0, 0.7, uv.shape)
    from apex_solver_amd.synthetic import BAProblemData

    return BAProblemData(base.poses, base.intr, base.points, cam_idx, pt_idx, np.ascontiguousarray(uv))


@pytest.mark.parametrize("mode", ["selfcal", "ba"])
def test_ragged_landmarks(oracle, mode):
    """k = 0 (unobserved landmark), k = 1, a duplicated camera, k = 64/65/129/200 (block-split
    landmarks use the off-diagonal scatter tasks), mixed in one problem."""
    n_cam = 320
    rng = np.random.default_rng(9)
    # range(300): more partners than one batch of the first row kernels (256) and more neighbours than one LDS
    # chunk (112 cameras at 9 DOF) -> split batches and several row tasks per camera
    lists = [[], [3], [5, 5, 9], list(range(64)), list(range(65)), list(range(40, 169)), list(range(200)),
             [7, 8, 7, 8, 100], list(range(300)), [319, 0, 319]]
    lists += [sorted(rng.choice(n_cam, size=int(rng.integers(2, 12)), replace=False).tolist()) for _ in range(300)]
    d = _custom(n_cam, len(lists), lists)
    prob, s = gpu_solver(d, mode)
    o = oracle_problem(oracle, d, prob, mode)
    assert s.compute_cost() == pytest.approx(o.residuals()[0], rel=1e-13)
    o.linearize()
    ostep, ograd, oS, ogred = o.solve_augmented(1e-3, 0, want_schur=True)
    step = s.solve_augmented_equation(1e-3)
    S, gred = s.get_schur()
    errs = dict(r=rel(s.get_residual(), o.residuals()[1]), grad=rel(s.get_gradient(), ograd), S=rel(S, oS),
                gred=rel(gred, ogred), step=rel(step, ostep))
    print(mode, {k: f"{v:.1e}" for k, v in errs.items()}, s.info())
    assert errs["r"] < 1e-12 and errs["grad"] < 1e-12 and errs["S"] < 1e-12 and errs["gred"] < 1e-10
    nc = prob.layout.cam_dof
    bwd = np.linalg.norm(oS @ step[:nc] - ogred) / (referee.sym_norm2(oS) * np.linalg.norm(step[:nc]) + np.linalg.norm(ogred))
    assert bwd < 1e-13 and errs["step"] < STEP_FORWARD_BOUND, (bwd, errs["step"])
    dc = 9 if mode == "selfcal" else 6
    referee.check_step(o, s, step, ostep, 1e-3, dc, label=f"ragged {mode}")
    referee.check_well_conditioned(o, s, dc, label=f"ragged {mode}")
    s.close()


@pytest.mark.parametrize("mode", ["selfcal", "ba"])
def test_the_two_layouts_of_the_pair_list_agree(mode):
    """The two layouts of the sorted pair list -- queued (4, the default with nine columns per camera: every lane group owns a
    block) and the first one (3: seven groups share a running block and fold; what six-column cameras run, where asking for 4
    arrives at 3) -- build the same S, g_red and gradient, also on landmarks with more than 64 partners per observation, and
    camera pairs with more common landmarks than one chunk of the pair list.  (The fused pair kernels, the LDS row forms and
    the global-atomics form of rounds 1-3 are deleted; their switch values are refused.)"""
    rng = np.random.default_rng(5)
    base = pkg.synthetic.make_problem(150, 6000, 3, 9, config_id=61)
    lists = [sorted(rng.choice(150, size=int(k), replace=False).tolist()) for k in rng.integers(2, 9, size=500)]
    lists += [list(range(150)), list(range(0, 150, 2)), list(range(140))]       # 150 / 75 / 140 observations
    wide = _custom(150, len(lists), lists)
    # two cameras that share 2,600 landmarks (a block of more than four 576-pair pieces in the queued layout: atomic adds),
    # with a third camera on a quarter of them, a camera that sees a landmark twice, and rows of one or two pairs
    heavy_lists = [[3, 7] + ([int(rng.integers(8, 20))] if l % 4 == 0 else []) for l in range(2600)]
    heavy_lists += [[5, 5, 9], [0, 1], [1, 2, 19]]
    heavy = _custom(20, len(heavy_lists), heavy_lists, seed=8)
    for d in (base, wide, heavy):
        out = []
        for form in (4, 3):
            ot = OptimizationType.SelfCalibration if mode == "selfcal" else OptimizationType.BundleAdjustment
            prob = Problem.bundle_adjustment(d, ot, 1.0)
            s = GpuSchurComplementSolver(0).with_option("schur_form", form).initialize_structure(prob)
            s.set_parameters(d.poses, d.intr, d.points)
            step = s.solve_augmented_equation(1e-3)
            S, gred = s.get_schur()
            out.append((S, gred, s.get_gradient(), step))
            s.close()
        assert rel(out[0][0], out[1][0]) < 1e-13 and rel(out[0][1], out[1][1]) < 1e-12
        assert rel(out[0][2], out[1][2]) < 1e-13
    with pytest.raises(Exception):
        GpuSchurComplementSolver(0).with_option("schur_form", 2).initialize_structure(prob)   # the LDS row form is gone


@pytest.mark.parametrize("mode", ["selfcal", "ba"])
def test_nested_dissection_ordering_matches_natural_and_oracle(oracle, mode):
    """The internal camera permutation (nested dissection of the tile graph, level-scheduled
    factorisation) is invisible at the boundary: same S, g_red and step as the caller's order and as
    the oracle."""
    d = pkg.synthetic.make_problem(560, 12000, 3, 8, config_id=71)
    res = {}
    for nd in (1, 0, 4):
        ot = OptimizationType.SelfCalibration if mode == "selfcal" else OptimizationType.BundleAdjustment
        prob = Problem.bundle_adjustment(d, ot, 1.0)
        s = GpuSchurComplementSolver(0).with_option("nested_dissection", nd).initialize_structure(prob)
        s.set_parameters(d.poses, d.intr, d.points)
        step = s.solve_augmented_equation(1e-3)
        S, gred = s.get_schur()
        res[nd] = (S, gred, step, s.get_gradient(), s.info())
        trial = s.eval_step(); s.commit_step()
        p2 = s.get_parameters()
        res[nd] += (trial, p2)
        s.close()
    print({k: (v[4]["etree_levels"], v[4]["tiles"]) for k, v in res.items()})
    assert res[1][4]["etree_levels"] < res[0][4]["etree_levels"]       # the chain became a tree
    for nd in (1, 4):
        assert rel(res[nd][0], res[0][0]) < 1e-12 and rel(res[nd][1], res[0][1]) < 1e-11
        assert rel(res[nd][3], res[0][3]) < 1e-13
        assert rel(res[nd][2], res[0][2]) < 1e-6
        assert res[nd][5] == pytest.approx(res[0][5], rel=1e-9)
        assert rel(res[nd][6][0], res[0][6][0]) < 1e-9 and rel(res[nd][6][2], res[0][6][2]) < 1e-8
    o = oracle_problem(oracle, d, prob, mode)
    o.linearize()
    ostep, ograd, oS, ogred = o.solve_augmented(1e-3, 0, want_schur=True)
    assert rel(res[1][0], oS) < 1e-12 and rel(res[1][1], ogred) < 1e-10 and rel(res[1][3], ograd) < 1e-12
    nc = prob.layout.cam_dof
    step = res[1][2]
    bwd = np.linalg.norm(oS @ step[:nc] - ogred) / (referee.sym_norm2(oS) * np.linalg.norm(step[:nc]) + np.linalg.norm(ogred))
    assert bwd < 1e-13


def test_cheirality_and_no_loss(oracle):
    """Points behind a camera give zero residual/Jacobian (projection_factor.rs:227-238); without a
    loss function the corrector is skipped (linearizer/mod.rs:144)."""
    d = pkg.synthetic.make_problem(12, 400, 3, 6, config_id=102, behind_frac=0.05)
    for huber in (1.0, None):
        prob, s = gpu_solver(d, "selfcal", huber=huber)
        o = oracle_problem(oracle, d, prob, "selfcal")
        c, r = o.residuals()
        assert np.sum((r.reshape(-1, 2) == 0).all(1)) > 0
        assert s.compute_cost() == pytest.approx(c, rel=1e-13)
        assert rel(s.get_residual(), r) < 1e-12
        _, _, oJp, oJl, oJi = o.linearize()
        ostep, ograd, oS, ogred = o.solve_augmented(1e-3, 0, want_schur=True)
        step = s.solve_augmented_equation(1e-3)
        S, gred = s.get_schur()
        scale = np.abs(oS).max()
        print("huber", huber, rel(S, oS), rel(step, ostep))
        assert rel(s.get_gradient(), ograd) < 1e-12
        # a landmark almost on a camera centre makes S = Hcc - E cancel terms ~1e12 (see
        # tests/test_oracle_golden.py); compare against the size of what is cancelled
        hinv, gl = s.get_landmark_blocks()
        assert np.isfinite(S).all() and np.isfinite(step).all()
        assert np.abs(S - oS).max() / scale < 1e-6
        # ... and camera pair by camera pair, against the magnitude of the terms that make up THAT block:
        #   S_ij = Hcc_ij - sum_l W_il Hll_l^-1 W_jl^T,   rounding <= c eps (|Hcc_ij| + sum_l |W_il| |Hll_l^-1| |W_jl|)
        # (the triple product cancels inside itself when Hll is nearly singular, so |E_ij| alone would understate it: with no
        # loss function the near-singular landmark of this problem gives 2e-10 relative to |E_ij| in the oracle as well).  A pair of
        # cameras that does not see that landmark must then match to ~1e-13 of ITS OWN terms however large the blocks next to
        # it are -- the global bound above would let it be wrong by 1e-6 * 1e11.
        lay = prob.layout
        n_cam = d.n_cam
        Jc_all = np.concatenate([oJi.reshape(-1, 2, 3), oJp.reshape(-1, 2, 6)], axis=2)      # (n_obs, 2, 9), columns [intr | pose]
        W = np.einsum("kri,krj->kij", Jc_all, oJl.reshape(-1, 2, 3))                          # W_k = Jc_k^T Jl_k  (9 x 3)
        wn = np.linalg.norm(W, ord=2, axis=(1, 2))
        hn = np.linalg.norm(hinv, ord=2, axis=(1, 2))
        bound = np.zeros((n_cam, n_cam))
        for c, k in zip(d.cam_idx, range(len(d.cam_idx))):
            bound[c, c] += np.linalg.norm(Jc_all[k], ord=2) ** 2                              # |Hcc| of the camera's own block
        order = np.argsort(d.pt_idx, kind="stable")
        ptr = np.searchsorted(d.pt_idx[order], np.arange(d.n_pt + 1))
        for l in range(d.n_pt):
            ks = order[ptr[l]:ptr[l + 1]]
            if len(ks):
                cams = d.cam_idx[ks]
                bound[np.ix_(cams, cams)] += hn[l] * np.outer(wn[ks], wn[ks])
        def cam_of(col):   # reference columns: [intr_c (3) ... | pose_c (6) ...]; n_cam <= 10^4: lexicographic = numeric order
            return col // 3 if col < 3 * n_cam else (col - 3 * n_cam) // 6
        cam_cols = np.array([cam_of(c) for c in range(lay.cam_dof)])
        err = np.zeros((n_cam, n_cam))
        np.maximum.at(err, (cam_cols[:, None].repeat(lay.cam_dof, 1), cam_cols[None, :].repeat(lay.cam_dof, 0)), np.abs(S - oS))
        seen = bound > 0
        assert (err[~seen] == 0).all()
        worst = float((err[seen] / bound[seen]).max())
        print("huber", huber, "worst camera-pair error relative to the magnitude of that pair's own terms", worst,
              "pair magnitudes", float(bound[seen].min()), float(bound.max()))
        assert worst < 1e-12
        # the step itself: against the exact step of this linearisation, next to the fp64 oracle's error, and at
        # lambda = 1e4 against the fp64 oracle outright
        referee.check_step(o, s, step, ostep, 1e-3, 9, label=f"cheirality huber={huber}")
        referee.check_well_conditioned(o, s, 9, label=f"cheirality huber={huber}")
        s.close()


def test_rejected_step_round_trip(oracle):
    """A rejected step is undone by the inverse retraction (optimizer/mod.rs:343-356)."""
    d = pkg.synthetic.make_problem(10, 300, 3, 6, config_id=8)
    prob, s = gpu_solver(d, "selfcal")
    o = oracle_problem(oracle, d, prob, "selfcal")
    o.linearize()
    ostep, _ = o.solve_augmented(1e-3, 0)
    s.solve_augmented_equation(1e-3)
    s.eval_step(); s.discard_step()
    o.apply_step(ostep, 1.0); o.apply_step(ostep, -1.0)
    po, io, lo = o.get_params(); pg, ig, lg = s.get_parameters()
    assert rel(pg, po) < 1e-12 and rel(lg, lo) < 1e-12 and rel(ig, io) < 1e-12
    assert np.abs(lg - d.points).max() < 1e-12  # back where it started up to rounding
    s.close()


def test_error_behaviour():
    """Error classes mirror LinAlgError (src/linalg/mod.rs:76-101)."""
    from apex_solver_amd.capi import LinAlgError

    d = pkg.synthetic.make_problem(6, 40, 3, 5, config_id=101)
    s = GpuSchurComplementSolver(0)
    with pytest.raises(LinAlgError) as e:  # solve before initialize_structure (explicit_schur.rs:1138-1142)
        s.solve_augmented_equation(1e-3)
    assert e.value.kind == "InvalidInput"
    prob = Problem.bundle_adjustment(d)
    s.initialize_structure(prob)
    with pytest.raises(LinAlgError) as e:  # parameters not set
        s.solve_augmented_equation(1e-3)
    assert e.value.kind == "InvalidState"
    assert s.get_gradient() is None  # LM turns this into NumericalInstability (levenberg_marquardt.rs:743-745)
    bad = pkg.synthetic.make_problem(6, 40, 3, 5, config_id=101)
    bad.cam_idx = bad.cam_idx.copy(); bad.cam_idx[3] = 99
    with pytest.raises(LinAlgError) as e:
        GpuSchurComplementSolver(0).initialize_structure(Problem.bundle_adjustment(bad))
    assert e.value.kind == "InvalidInput"
    s.close()


def test_lm_converges_like_reference_integration_test(oracle):
    """tests/bundle_adjustment_integration.rs:33-153 asserts only: converged status,
    final_cost < initial_cost, RMSE decreased.  Same assertions, plus agreement with the oracle."""
    d = pkg.synthetic.make_problem(21, 1100, 3, 8, config_id=21)
    prob = Problem.bundle_adjustment(d, OptimizationType.SelfCalibration, 1.0)
    cfg = LevenbergMarquardtConfig.for_bundle_adjustment().with_schur_variant(SchurVariant.Sparse).with_max_iterations(50)
    res = LevenbergMarquardt.with_config(cfg).optimize(prob)
    assert res.status in (OptimizationStatus.CostToleranceReached, OptimizationStatus.ParameterToleranceReached,
                          OptimizationStatus.GradientToleranceReached, OptimizationStatus.MaxIterationsReached)
    assert res.final_cost < res.initial_cost
    o = oracle_problem(oracle, d, prob, "selfcal")
    ores = o.optimize(oracle.LMConfig.default(max_iterations=50, variant=0))
    print(res.status, res.iterations, res.final_cost, "| oracle", ores.status, ores.iterations, ores.final_cost)
    assert res.status.name == ores.status and res.iterations == ores.iterations
    assert res.final_cost == pytest.approx(ores.final_cost, rel=1e-6)


def test_bal_file_end_to_end(oracle, tmp_path):
    """A BAL text file drives the backend: C++ reader -> problem as
    bin/bundle_adjustment.rs builds it -> LevenbergMarquardt; same trajectory as the oracle on the same
    loaded data (the reference's integration test asserts: converged, cost decreased)."""
    from apex_solver_amd.bal import BalLoader, write_bal

    d0 = pkg.synthetic.make_problem(15, 500, 3, 7, config_id=33)
    path = tmp_path / "problem-15-500-pre.txt"
    write_bal(path, d0)
    d = BalLoader.load(path).to_problem_data()
    prob = Problem.bundle_adjustment(d, OptimizationType.SelfCalibration, 1.0)
    cfg = LevenbergMarquardtConfig.for_bundle_adjustment().with_schur_variant(SchurVariant.Sparse).with_max_iterations(10)
    res = LevenbergMarquardt.with_config(cfg).optimize(prob)
    assert res.final_cost < res.initial_cost
    o = oracle_problem(oracle, d, prob, "selfcal")
    ores = o.optimize(oracle.LMConfig.default(max_iterations=10, variant=0))
    assert res.iterations == ores.iterations and res.status.name == ores.status
    assert res.initial_cost == pytest.approx(ores.initial_cost, rel=1e-13)
    assert res.final_cost == pytest.approx(ores.final_cost, rel=1e-6)


def test_rccl_communicator_single_rank():
    """The collective code path (RCCL communicator, all-reduces on the solver's stream) with a
    one-rank communicator gives the same step as the plain path.  (N > 1 needs N GPUs: the driver's
    scaling run; the exchange arithmetic is covered by test_shard_partials_sum_to_full and, on CPU,
    tests/test_multirank_gloo.py.)"""
    import ctypes as C

    d = pkg.synthetic.make_problem(40, 2000, 3, 7, config_id=240)
    prob, s0 = gpu_solver(d, "selfcal")
    step0 = s0.solve_augmented_equation(1e-3)
    trial0 = s0.eval_step()
    buf = (C.c_char * 128)()
    assert pkg.capi.load().apexgpu_get_unique_id(C.cast(buf, C.c_void_p)) == 0
    s1 = GpuSchurComplementSolver(0).with_communicator(1, 0, bytes(buf))
    s1.initialize_structure(prob)
    s1.set_parameters(d.poses, d.intr, d.points)
    step1 = s1.solve_augmented_equation(1e-3)
    assert rel(step1, step0) < 1e-9
    assert s1.step_stats()[0] == pytest.approx(s0.step_stats()[0], rel=1e-12)
    assert s1.eval_step() == pytest.approx(trial0, rel=1e-10)
    s1.commit_step()
    p1 = s1.get_parameters()
    s0.commit_step()
    assert rel(p1[2], s0.get_parameters()[2]) < 1e-9
    s0.close(); s1.close()


def test_shard_partials_sum_to_full(oracle):
    """Landmark shards (no communicator): partial S and g_red of the two halves add up to the full
    ones -- the quantity the RCCL all-reduce sums."""
    d = pkg.synthetic.make_problem(40, 2000, 3, 7, config_id=240)
    prob, s = gpu_solver(d, "selfcal")
    s.solve_augmented_equation(1e-3)
    S, gred = s.get_schur()
    parts = []
    for r in range(2):
        _, sr = gpu_solver(d, "selfcal", shard=(r, 2))
        sr.assemble(1e-3)
        parts.append(sr.get_schur())
        sr.close()
    S2 = parts[0][0] + parts[1][0]; g2 = parts[0][1] + parts[1][1]
    assert rel(S2, S) < 1e-12 and rel(g2, gred) < 1e-11
    # the matrix-free operator shards the same way: every S p is a sum of the ranks' partial products
    # (what the implicit variant all-reduces once per PCG iteration)
    x = np.random.default_rng(3).normal(size=prob.layout.cam_dof)
    ye, yi = s.schur_matvec(1e-3, x)
    pe = np.zeros_like(ye); pi = np.zeros_like(yi)
    for r in range(3):
        _, sr = gpu_solver(d, "selfcal", shard=(r, 3))
        a, b = sr.schur_matvec(1e-3, x)
        pe += a; pi += b
        sr.close()
    assert rel(pe, ye) < 1e-12 and rel(pi, yi) < 1e-12 and rel(yi, ye) < 1e-12
    s.close()


# ------------------------------------------------------------------------------------------------
# Jacobi column scaling (optimizer/mod.rs:749-763; linearizer/mod.rs:229-262)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["selfcal", "ba"])
def test_jacobi_scaling_one_iteration_vs_oracle(oracle, mode):
    """Column norms, the scaled Schur system (S, g_red), the scaled step / gradient the solver returns, the
    statistics compute_step_generic derives from them, and the trial cost of the unscaled step."""
    d = pkg.synthetic.make_problem(40, 2000, 3, 7, config_id=240)
    prob, s = gpu_solver(d, mode)
    o = oracle_problem(oracle, d, prob, mode)
    o.linearize()
    norms = s.compute_column_norms()
    onorms = o.column_norms()
    assert rel(norms, onorms) < 1e-12
    scal = 1.0 / (1.0 + onorms)
    s.apply_column_scaling(scal)
    o.set_column_scaling(scal)
    lam = 1e-3
    oy, ogs, oS, ogred = o.solve_augmented(lam, 0, want_schur=True)
    y = s.solve_augmented_equation(lam)
    S, gred = s.get_schur()
    errs = dict(grad=rel(s.get_gradient(), ogs), S=rel(S, oS), gred=rel(gred, ogred), step=rel(y, oy))
    print(mode, {k: f"{v:.1e}" for k, v in errs.items()})
    assert errs["grad"] < 1e-12 and errs["S"] < 1e-12 and errs["gred"] < 1e-10
    tol = STEP_FORWARD_BOUND
    nc = prob.layout.cam_dof
    bwd = np.linalg.norm(oS @ y[:nc] - ogred) / (referee.sym_norm2(oS) * np.linalg.norm(y[:nc]) + np.linalg.norm(ogred))
    assert bwd < 1e-13 and errs["step"] < tol, (bwd, errs["step"], tol)
    # referee on the SCALED system (the oracle scales its blocks in fp64 exactly as apply_column_scaling does; the device's
    # exported blocks are unscaled, so only the oracle's linearisation is refereed here)
    referee.check_step(o, s, y, oy, lam, 9 if mode == "selfcal" else 6, label=f"jacobi scaling {mode}", own=False)
    # compute_step_generic (levenberg_marquardt.rs:746-760): |scaled gradient|, |unscaled step|,
    # predicted reduction from the unscaled step and the scaled gradient
    ostep = oy * scal
    gn, sn, pred = s.step_stats()
    assert gn == pytest.approx(np.linalg.norm(ogs), rel=1e-12)
    assert sn == pytest.approx(np.linalg.norm(ostep), rel=1e-8)
    assert pred == pytest.approx(0.5 * ostep @ (lam * ostep - ogs), rel=1e-7)
    assert rel(s.apply_inverse_scaling(y), ostep) < tol
    o.apply_step(ostep, 1.0)
    assert s.eval_step() == pytest.approx(o.residuals()[0], rel=1e-9)
    s.discard_step()
    # switching the scaling off restores the plain solve
    s.apply_column_scaling(None)
    o.apply_step(ostep, -1.0); o.set_column_scaling(None); o.linearize()
    ostep0, ograd0 = o.solve_augmented(lam, 0)
    step0 = s.solve_augmented_equation(lam)
    assert rel(s.get_gradient(), ograd0) < 1e-11 and rel(step0, ostep0) < 1e-7
    s.close()


@pytest.mark.parametrize("variant", [SchurVariant.Iterative, SchurVariant.Implicit])
def test_jacobi_scaling_pcg_variants_vs_oracle(oracle, variant):
    """Both PCG variants iterate on the scaled system (tolerance on the scaled residual)."""
    d = pkg.synthetic.make_problem(30, 1500, 3, 7, config_id=241)
    prob, s = gpu_solver(d, "selfcal", variant=variant)
    o = oracle_problem(oracle, d, prob, "selfcal")
    if variant == SchurVariant.Implicit:
        s.with_cg_params(500, 1e-9); o.set_cg_params(500, 1e-9)
    o.linearize()
    scal = 1.0 / (1.0 + o.column_norms())
    s.apply_column_scaling(scal); o.set_column_scaling(scal)
    oy_chol, ogs, oS, ogred = o.solve_augmented(1e-3, 0, want_schur=True)
    oy, _ = o.solve_augmented(1e-3, variant.value)
    y = s.solve_augmented_equation(1e-3)
    nc = prob.layout.cam_dof
    it_gpu, it_ora = s.info()["pcg_iterations"], o.last_pcg_iters
    print(variant.name, "pcg iterations gpu/oracle", it_gpu, it_ora)
    assert abs(it_gpu - it_ora) <= max(3, it_ora // 20)
    assert rel(s.get_gradient(), ogs) < 1e-12
    tol = 1e-6 if variant == SchurVariant.Iterative else 1e-9
    r_gpu = np.linalg.norm(oS @ y[:nc] - ogred); r_ora = np.linalg.norm(oS @ oy[:nc] - ogred)
    assert r_gpu < 10 * max(r_ora, tol * max(np.linalg.norm(ogred), 1.0))
    if variant == SchurVariant.Implicit:  # both operator forms agree in the scaled variables too
        x = np.random.default_rng(5).standard_normal(nc)
        ye, yi = s.schur_matvec(1e-3, x)
        assert rel(ye, oS @ x) < 1e-11 and rel(yi, oS @ x) < 1e-11
    s.close()


def test_jacobi_scaling_lm_history_vs_oracle(oracle):
    """LevenbergMarquardtConfig::with_jacobi_scaling(true): scaling from the Jacobian of iteration 0, same
    accept / reject pattern, costs and damping as the oracle's loop."""
    d = pkg.synthetic.make_problem(20, 600, 3, 7, config_id=242)
    prob = Problem.bundle_adjustment(d, OptimizationType.SelfCalibration, 1.0)
    cfg = LevenbergMarquardtConfig().with_max_iterations(10).with_jacobi_scaling(True)
    res = LevenbergMarquardt.with_config(cfg).optimize(prob)
    o = oracle.from_data(d, prob.layout, mode="selfcal")
    ores = o.optimize(oracle.LMConfig.default(max_iterations=10, use_jacobi_scaling=1))
    assert res.status.name == ores.status and res.iterations == ores.iterations
    assert np.array_equal(res.history[:, 3], ores.history[:, 3])
    assert np.allclose(res.history[:, 0], ores.history[:, 0], rtol=1e-7)
    assert np.allclose(res.history[:, 4], ores.history[:, 4], rtol=1e-6)   # |scaled gradient|
    assert np.allclose(res.history[:, 1], ores.history[:, 1], rtol=1e-4)
    # and the unscaled loop takes a different path from the same start
    res0 = LevenbergMarquardt.with_config(cfg.with_jacobi_scaling(False)).optimize(prob)
    assert not np.allclose(res0.history[:2, 5], res.history[:2, 5], rtol=1e-3)


@pytest.mark.parametrize("mode", ["selfcal", "ba"])
def test_get_hessian_matches_jtj_of_the_exported_blocks(mode):
    """LinearSolver::get_hessian (explicit_schur.rs:1146-1160, 1236-1238): H = J^T J, undamped, full symmetric CSC in the
    global column order -- against scipy's product of the Jacobian rebuilt from the exported blocks; duplicated
    (camera, landmark) factors, unobserved variables and points behind a camera included."""
    lists = [[], [3], [5, 5, 9], [7, 8, 7, 8, 10], [11, 0, 11]]
    rng = np.random.default_rng(4)
    lists += [sorted(rng.choice(12, size=int(rng.integers(2, 7)), replace=False).tolist()) for _ in range(120)]
    d = _custom(12, len(lists), lists)
    prob, s = gpu_solver(d, mode)
    dc = 9 if mode == "selfcal" else 6
    s.solve_augmented_equation(1e-3)
    H = s.get_hessian()
    jc, jl = s.get_jacobian_blocks()
    jp, ji = jc_to_blocks(jc, dc)
    J = np_ref.sparse_jacobian(jp, jl, ji, d.cam_idx, d.pt_idx, prob.layout, selfcal=(mode == "selfcal"))
    ref = (J.T @ J).tocsc()
    assert H.shape == ref.shape == (prob.total_dof, prob.total_dof)
    diff = (H - ref)
    assert abs(diff).max() <= 1e-12 * abs(ref).max()
    assert abs(H - H.T).max() == 0.0
    H.sort_indices()
    assert np.all(np.diff(H.indptr) >= 0) and H.has_sorted_indices
    # gradient consistency: H is the matrix of the normal equations the step solves (up to lambda I)
    step = s.solve_augmented_equation(1e-3)
    g = s.get_gradient()
    res = H @ step + 1e-3 * step + g
    assert np.linalg.norm(res) <= 1e-9 * (np.linalg.norm(g) + 1.0)
    if mode == "ba":   # no factor touches the intrinsics: their columns are structurally empty
        lay = prob.layout
        for c in range(d.n_cam):
            assert H[:, lay.intr_col[c]:lay.intr_col[c] + 3].nnz == 0
    s.close()


@pytest.mark.parametrize("mode", ["selfcal", "ba"])
def test_record_form_of_the_back_substitution(oracle, mode):
    """k_back_substitute<.., REC>: the landmark steps come from the projection records k_landmark_reduce wrote for the same
    linearisation; explicit and matrix-free Schur variants (the matrix-free operator's landmark half is the same kernel).
    The records must be those of the NEW parameters after a committed step: the same solve on a fresh handle at these
    parameters (stale records would be off by ~1e-3); the step itself is held to the oracle by the parity cases above."""
    d = pkg.synthetic.make_problem(14, 900, 3, 8, config_id=33)
    for variant in (SchurVariant.Sparse, SchurVariant.Iterative):
        ot = OptimizationType.SelfCalibration if mode == "selfcal" else OptimizationType.BundleAdjustment
        prob = Problem.bundle_adjustment(d, ot, 1.0)
        s = GpuSchurComplementSolver(0).with_variant(variant)
        s.initialize_structure(prob)
        s.set_parameters(d.poses, d.intr, d.points)
        s.solve_augmented_equation(1e-3)
        s.eval_step(); s.commit_step()
        after = s.solve_augmented_equation(1e-3)
        params = s.get_parameters()
        s.close()
        s = GpuSchurComplementSolver(0).with_variant(variant)
        s.initialize_structure(prob)
        s.set_parameters(*params)
        # (not bitwise: blocks of S shared by several waves are summed with atomics)
        assert rel(after, s.solve_augmented_equation(1e-3)) < 1e-7
        s.close()


def test_schur_assembly_is_reproducible_bit_for_bit():
    """With the queued pair layout every block of S is summed by one lane group in list order and stored once (a block cut
    between two queues of a task is joined in registers, in a fixed order); the camera and landmark passes fold in fixed
    orders too.  Two assemblies of the same linearisation are therefore the same bits -- on a shape with several tile rows,
    split tasks and nested-dissection levels, but no block beyond 576 pairs and no camera that sees a landmark twice (those
    add atomically)."""
    d = pkg.synthetic.make_problem(420, 12000, 3, 8, config_id=91)
    prob = Problem.bundle_adjustment(d, OptimizationType.SelfCalibration, 1.0)
    s = GpuSchurComplementSolver(0).initialize_structure(prob)
    s.set_parameters(d.poses, d.intr, d.points)
    s.assemble(1e-3)
    S1, g1 = s.get_schur()
    S1, g1 = S1.copy(), g1.copy()
    s.assemble(1e-3)
    S2, g2 = s.get_schur()
    assert np.array_equal(S1, S2) and np.array_equal(g1, g2)
    a = s.solve_augmented_equation(1e-3).copy()
    b = s.solve_augmented_equation(1e-3).copy()
    assert np.array_equal(a, b)
    s.close()


# ---- round 5: the records of the pair list written by the device ---------------------------------------------------------------
def test_device_built_pair_list_is_the_host_list():
    """The records of the queued pair list are written by the device from the observation lists ("device_pair_list", default;
    k_build_pair_recs_q) instead of being built on the host and copied.  On a banded problem the two lists are the same slot for
    slot and S the same bit for bit; so are they with cameras that see a landmark twice and landmarks of 64..300 observations
    (split blocks, pieces; S then agrees to rounding: such blocks add atomically), and for six-column cameras."""
    PAD = np.uint32(0xFFFFFFFF)

    def build(d, dev):
        prob = Problem.bundle_adjustment(d, OptimizationType.SelfCalibration, 1.0)
        s = GpuSchurComplementSolver(0).with_option("device_pair_list", dev)
        s.initialize_structure(prob)
        s.set_parameters(d.poses, d.intr, d.points)
        recs = s.pair_records()
        s.assemble(1e-3)
        S, g = s.get_schur()
        s.close()
        return recs, S, g

    d = pkg.synthetic.make_problem(130, 4000, 3, 9, config_id=515)
    rh, Sh, gh = build(d, 0)
    rd, Sd, gd = build(d, 1)
    assert rh.shape == rd.shape and rh.shape[0] % 64 == 0
    real_h, real_d = rh[:, 0] != PAD, rd[:, 0] != PAD
    assert np.array_equal(real_h, real_d) and real_h.sum() > 10000
    assert np.array_equal(rh[real_h], rd[real_d])
    assert np.array_equal(Sh, Sd) and np.array_equal(gh, gd)

    n_cam = 320
    rng = np.random.default_rng(9)
    lists = [[], [3], [5, 5, 9], list(range(64)), list(range(65)), list(range(40, 169)), list(range(200)),
             [7, 8, 7, 8, 100], list(range(300)), [319, 0, 319]]
    lists += [sorted(rng.choice(n_cam, size=int(rng.integers(2, 12)), replace=False).tolist()) for _ in range(300)]
    d = _custom(n_cam, len(lists), lists)
    rh, Sh, gh = build(d, 0)
    rd, Sd, gd = build(d, 1)
    real_h, real_d = rh[:, 0] != PAD, rd[:, 0] != PAD
    assert rh.shape == rd.shape and np.array_equal(real_h, real_d)
    assert np.array_equal(rh[real_h], rd[real_d])          # also with duplicated cameras, split blocks and pieces
    assert rel(Sd, Sh) < 1e-13 and rel(gd, gh) < 1e-13      # (blocks that add atomically: the order of the adds is not fixed)
    # six-column cameras (BundleAdjustment mode) run form 3 whatever is asked
    prob = Problem.bundle_adjustment(d, OptimizationType.BundleAdjustment, 1.0)
    s3 = GpuSchurComplementSolver(0).with_option("schur_form", 4)
    s3.initialize_structure(prob); s3.set_parameters(d.poses, d.intr, d.points)
    assert s3.info()["schur_form"] == 3
    s3.close()


# ---- round 6: decisions of a handle's life that must not stick, and state a new solve must void ---------------------------------
def test_a_second_structure_on_one_handle_is_decided_afresh(oracle):
    """the automatic matrix-free selection of one set_structure must not survive into the next one on the same
    handle.  Structure A (every landmark seen by cameras from all over: S dense at tile granularity) is refused at a lowered
    plan limit and answered matrix-free; structure B (the same counts, banded) fits: variant 0 is the Cholesky again -- S exists,
    the step is the oracle's -- and back to A the handle is matrix-free again."""
    n_cam, n_pt, k = 160, 1200, 4
    rng = np.random.default_rng(3)
    dense = [sorted(rng.choice(n_cam, size=k, replace=False).tolist()) for _ in range(n_pt)]
    banded = [sorted(((l * n_cam) // n_pt + rng.choice(12, size=k, replace=False)) % n_cam) for l in range(n_pt)]
    dA, dB = _custom(n_cam, n_pt, dense, seed=11), _custom(n_cam, n_pt, banded, seed=11)
    assert dA.n_obs == dB.n_obs
    probA = Problem.bundle_adjustment(dA, OptimizationType.SelfCalibration, 1.0)
    probB = Problem.bundle_adjustment(dB, OptimizationType.SelfCalibration, 1.0)
    s = GpuSchurComplementSolver(0).with_option("max_tile_updates", 60).with_option("variant_cost_permille", 0)
    s.initialize_structure(probA)
    assert s.variant_info()["variant_used"] == "Implicit" and s.variant_info()["variant_choice"] == "matrix-free: plan refused"
    s.set_parameters(dA.poses, dA.intr, dA.points)
    s.solve_augmented_equation(1e-3)
    assert s.info()["pcg_iterations"] > 0
    s.reinitialize_structure(probB)
    vi = s.variant_info()
    assert vi["variant_used"] == "Sparse" and vi["reason"] == "" and vi["variant_choice"] == "direct", vi
    s.set_parameters(dB.poses, dB.intr, dB.points)
    step = s.solve_augmented_equation(1e4)
    assert s.info()["pcg_iterations"] == 0 and s.info()["tiles"] > s.info()["tile_rows"]
    o = oracle_problem(oracle, dB, probB, "selfcal")
    o.linearize()
    ostep, _, oS, _ = o.solve_augmented(1e4, 0, want_schur=True)
    assert rel(step, ostep) < 1e-10
    S, _ = s.get_schur()
    assert rel(S, oS) < 1e-12
    s.reinitialize_structure(probA)
    assert s.variant_info()["variant_used"] == "Implicit"
    s.close()


def test_a_new_solve_voids_the_evaluated_trial_point():
    """with the eager step evaluation every solve overwrites the trial parameter set; a commit_step that follows
    eval_step -> solve_augmented must not silently take the NEW solve's trial point: it is refused until that step is evaluated."""
    d = pkg.synthetic.make_problem(20, 800, 3, 7, config_id=12)
    _, s = gpu_solver(d, "selfcal")
    s.solve_augmented_equation(1e-3)
    s.eval_step()
    s.solve_augmented_equation(1e-1)
    with pytest.raises(pkg.capi.LinAlgError):
        s.commit_step()
    c = s.eval_step()          # the step of the second solve, evaluated: now it can be committed
    s.commit_step()
    assert s.compute_cost() == pytest.approx(c, rel=1e-13)
    s.close()


def test_eager_step_evaluation_does_not_change_the_lm_history():
    """ "eager_step_eval" 0 (statistics, trial point and trial cost on request, three device round trips per LM iteration) against
    the default (enqueued behind the back-substitution, read at the solve's wait): the same LM history, cost for cost."""
    from apex_solver_amd.solver import LevenbergMarquardt, LevenbergMarquardtConfig
    d = pkg.synthetic.make_problem(30, 1500, 3, 7, config_id=14)
    hist = []
    for eager in (1, 0):
        prob, s = gpu_solver(d, "selfcal", options={"eager_step_eval": eager})
        res = LevenbergMarquardt.with_config(LevenbergMarquardtConfig().with_max_iterations(6)).optimize(prob, solver=s)
        hist.append((res.iterations, res.initial_cost, res.final_cost, np.asarray(res.history, dtype=float)))
        s.close()
    assert hist[0][0] == hist[1][0] and hist[0][1] == hist[1][1] and hist[0][2] == pytest.approx(hist[1][2], rel=1e-12), hist
    assert hist[0][3].shape == hist[1][3].shape and hist[0][3].size > 0
    assert np.allclose(hist[0][3], hist[1][3], rtol=1e-11, atol=0.0), np.abs(hist[0][3] - hist[1][3]).max()
