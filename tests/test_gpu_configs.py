"""Every BASELINE.json configuration at its NAMED size on the device (`pytest -m gpu`), plus an oracle comparison on a
<= 1/10 sample of the same generator.

  configs[2]  ladybug-1723   1,723 cameras / 156,502 landmarks / ~0.68 M observations, explicit Schur, 1 GPU
  configs[3]  venice-1778    1,778 / 993,923 / ~5.0 M, landmarks sharded over 8 ranks (lockstep on one GPU: the library
                             plays the all-reduces on exactly the buffers the RCCL path reduces), both sharding modes
  configs[4]  synthetic-10k  10,000 / 2,000,000 / 12 M, implicit-Schur matrix-free PCG
(configs[0] ladybug-49: tests/test_gpu_bench_contract.py; configs[1] sphere2500: tests/test_gpu_pg_parity.py; the headline
final-13682: tests/test_gpu_parity.py::test_full_size_explicit_and_matrix_free_schur_agree.)

At full size the oracle's dense S does not fit a test budget, so the checks are the size-independent properties of the
path: the explicit tiles and the matrix-free operator give the same S x (two code paths over all observations), S is
symmetric positive definite on probes, the returned camera step solves S dc = g_red to a normwise backward error of
1e-13, the full step solves the damped normal equations rebuilt from the exported Jacobian blocks (ladybug-1723), the
predicted reduction is positive and the cost goes down.
"""
import numpy as np
import pytest

import apex_solver_amd as pkg
import np_ref
import referee
from apex_solver_amd.solver import GpuSchurComplementSolver, OptimizationType, Problem, SchurVariant

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = np.ravel(a); b = np.ravel(b)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def make(d, mode="selfcal", variant=SchurVariant.Sparse, shard=None, opts=()):
    ot = OptimizationType.SelfCalibration if mode == "selfcal" else OptimizationType.BundleAdjustment
    prob = Problem.bundle_adjustment(d, ot, 1.0)
    s = GpuSchurComplementSolver(0).with_variant(variant)
    for k, v in opts:
        s.with_option(k, v)
    if shard:
        s.with_shard(*shard)
    s.initialize_structure(prob)
    s.set_parameters(d.poses, d.intr, d.points)
    return prob, s


def schur_norm2(s, lam, n, iters=12, seed=0):
    """|S|_2 by power iteration through the explicit tiles."""
    x = np.random.default_rng(seed).normal(size=n)
    nrm = 0.0
    for _ in range(iters):
        x /= np.linalg.norm(x)
        y, _ = s.schur_matvec(lam, x, implicit=False)
        nrm = float(np.linalg.norm(y))
        x = y
    return nrm


def camera_step_checks(s, prob, lam, step, tag):
    """S dc = g_red through the explicit tiles AND the matrix-free operator; returns the measured numbers."""
    nc = prob.layout.cam_dof
    _, gred = s.get_schur(want_S=False)
    dc = step[:nc]
    ye, yi = s.schur_matvec(lam, dc)
    Sn = schur_norm2(s, lam, nc)
    bwd_e = np.linalg.norm(ye - gred) / (Sn * np.linalg.norm(dc) + np.linalg.norm(gred))
    bwd_i = np.linalg.norm(yi - gred) / (Sn * np.linalg.norm(dc) + np.linalg.norm(gred))
    print(f"{tag}: |S|_2 ~ {Sn:.3e}  backward error explicit {bwd_e:.2e} / matrix-free {bwd_i:.2e}  "
          f"explicit vs matrix-free S.dc {rel(ye, yi):.2e}")
    assert rel(ye, yi) < 1e-10
    return bwd_e, bwd_i


def probe_symmetry(s, prob, lam):
    rng = np.random.default_rng(1)
    nc = prob.layout.cam_dof
    x = rng.normal(size=nc); y = rng.normal(size=nc)
    sx_e, sx_i = s.schur_matvec(lam, x)
    sy_e, _ = s.schur_matvec(lam, y, implicit=False)
    assert rel(sx_e, sx_i) < 1e-11
    assert abs(x @ sy_e - y @ sx_e) <= 1e-11 * abs(x @ sy_e)
    assert x @ sx_e > 0 and y @ sy_e > 0


def one_lm_iteration_behaves(s, lam):
    c0 = s.compute_cost()
    gn, sn, pred = s.step_stats()
    c1 = s.eval_step()
    assert pred > 0 and c1 < c0 and np.isfinite(gn) and np.isfinite(sn), (pred, c0, c1)
    s.discard_step()
    return c0, c1


def oracle_sample_check(oracle, shape, scale, mode, variant=0, cg=None):
    """The same generator at <= 1/10 of the named size against the C oracle: S, g_red, gradient <= 1e-12 / 1e-10,
    the step by backward error (<= 1e-13) and a fixed forward bound (1e-7)."""
    d = pkg.synthetic.make_named(shape, scale)
    gv = {0: SchurVariant.Sparse, 2: SchurVariant.Implicit}[variant]
    prob, s = make(d, mode, variant=gv)
    o = oracle.from_data(d, prob.layout, mode=mode, huber_delta=1.0)
    if cg:
        s.with_cg_params(*cg); o.set_cg_params(*cg)
    lam = 1e-3
    assert s.compute_cost() == pytest.approx(o.residuals()[0], rel=1e-13)
    o.linearize()
    ostep, ograd, oS, ogred = o.solve_augmented(lam, 0, want_schur=True)
    step = s.solve_augmented_equation(lam)
    nc = prob.layout.cam_dof
    if variant == 0:
        S, gred = s.get_schur()
        errs = dict(grad=rel(s.get_gradient(), ograd), S=rel(S, oS), gred=rel(gred, ogred), step=rel(step, ostep))
        bwd = np.linalg.norm(oS @ step[:nc] - ogred) / (referee.sym_norm2(oS) * np.linalg.norm(step[:nc]) + np.linalg.norm(ogred))
        print(f"{d.name} vs oracle:", {k: f"{v:.1e}" for k, v in errs.items()}, f"backward {bwd:.1e}")
        assert errs["grad"] < 1e-12 and errs["S"] < 1e-12 and errs["gred"] < 1e-10
        assert bwd < 1e-13 and errs["step"] < 1e-7
        referee.check_step(o, s, step, ostep, lam, 9 if mode == "selfcal" else 6, label=d.name)
    else:
        istep, _ = o.solve_augmented(lam, 2)
        it_g, it_o = s.info()["pcg_iterations"], o.last_pcg_iters
        r_gpu = np.linalg.norm(oS @ step[:nc] - ogred); r_ora = np.linalg.norm(oS @ istep[:nc] - ogred)
        print(f"{d.name} implicit PCG iterations gpu/oracle {it_g}/{it_o}  residual {r_gpu:.2e}/{r_ora:.2e}")
        assert rel(s.get_gradient(), ograd) < 1e-12
        # on this ill-conditioned shape both stop on the tolerance after a few hundred iterations; the count moves with
        # rounding (FMA contraction), the residual they stop at does not
        assert abs(it_g - it_o) <= max(3, it_o // 4)
        assert r_gpu < 10 * max(r_ora, cg[1] * max(np.linalg.norm(ogred), 1.0))
    o.apply_step(ostep if variant == 0 else istep, 1.0)
    # (two PCG runs that stop at the same tolerance still differ by that tolerance's worth of step)
    assert s.eval_step() == pytest.approx(o.residuals()[0], rel=1e-6 if variant == 0 else 1e-4)
    s.close()


# ---- configs[2]: ladybug-1723, explicit Schur on one GPU ---------------------------------------------------------
def test_ladybug_1723_full_size_explicit_schur():
    d = pkg.synthetic.make_named("ladybug-1723")
    assert (d.n_cam, d.n_pt) == (1723, 156502)
    lam = 1e-3
    prob, s = make(d, "selfcal")
    probe_symmetry(s, prob, lam)
    step = s.solve_augmented_equation(lam)
    grad = s.get_gradient()
    bwd_e, bwd_i = camera_step_checks(s, prob, lam, step, "ladybug-1723")
    assert bwd_e < 1e-13 and bwd_i < 1e-12
    # the full step against the damped normal equations rebuilt (scipy.sparse) from the exported blocks
    jc, jl = s.get_jacobian_blocks()
    r = s.get_residual()
    J = np_ref.sparse_jacobian(jc[:, :, :6], jl, jc[:, :, 6:9], d.cam_idx, d.pt_idx, prob.layout, selfcal=True)
    g = J.T @ r
    assert rel(grad, g) < 1e-12
    res = J.T @ (J @ step) + lam * step + g
    x = np.random.default_rng(2).normal(size=J.shape[1])
    for _ in range(15):
        x /= np.linalg.norm(x); x = J.T @ (J @ x) + lam * x
    bwd = np.linalg.norm(res) / (np.linalg.norm(x) * np.linalg.norm(step) + np.linalg.norm(g))
    print("ladybug-1723 normal-equation backward error", bwd, s.info())
    assert bwd < 1e-13
    s.solve_augmented_equation(lam, want_step=False)
    one_lm_iteration_behaves(s, lam)
    s.close()


def test_ladybug_1723_sample_vs_oracle(oracle):
    oracle_sample_check(oracle, "ladybug-1723", 0.1, "selfcal")
    oracle_sample_check(oracle, "ladybug-1723", 0.05, "ba")


# ---- configs[3]: venice-1778, landmarks sharded over 8 ranks --------------------------------------------------------
def test_venice_1778_full_size_single_gpu():
    d = pkg.synthetic.make_named("venice-1778")
    assert (d.n_cam, d.n_pt) == (1778, 993923)
    lam = 1e-3
    prob, s = make(d, "selfcal")
    probe_symmetry(s, prob, lam)
    step = s.solve_augmented_equation(lam)
    bwd_e, bwd_i = camera_step_checks(s, prob, lam, step, "venice-1778")
    assert bwd_e < 1e-13 and bwd_i < 1e-12
    s.solve_augmented_equation(lam, want_step=False)
    one_lm_iteration_behaves(s, lam)
    s.close()


@pytest.mark.parametrize("tree", [1, 0], ids=["tree-sharded", "range-sharded"])
def test_venice_1778_full_size_lockstep_8_ranks(tree):
    """BASELINE configs[3] as north_star words it -- 8 landmark shards, the exchange of S / g_red before the solve --
    driven in lockstep on one GPU, in both sharding modes, against the single-rank solve of the same system."""
    d = pkg.synthetic.make_named("venice-1778")
    lam = 1e-3
    world = 8
    prob, s1 = make(d, "selfcal")
    step1 = s1.solve_augmented_equation(lam)
    _, gred = s1.get_schur(want_S=False)
    nc = prob.layout.cam_dof
    ranks = [make(d, "selfcal", shard=(r, world), opts=(("tree_sharding", tree),))[1] for r in range(world)]
    GpuSchurComplementSolver.lockstep_solve(ranks, lam)
    infos = [s.info() for s in ranks]
    print("venice-1778 x8:", "top columns", infos[0]["dist_top_columns"], "observations per rank", [i["local_obs"] for i in infos])
    assert sum(i["local_obs"] for i in infos) == d.n_obs
    owned = np.stack([s.owned_landmarks() for s in ranks])
    assert (owned.sum(0) == 1).all()                                   # every landmark on exactly one rank
    steps = [s.export_step()[0] for s in ranks]
    for r in range(1, world):
        assert np.array_equal(steps[r][:nc], steps[0][:nc])            # bit-identical camera step on all ranks
    Sx, _ = s1.schur_matvec(lam, steps[0][:nc], implicit=False)
    S1, _ = s1.schur_matvec(lam, step1[:nc], implicit=False)
    r_dist = np.linalg.norm(Sx - gred) / np.linalg.norm(gred)
    r_one = np.linalg.norm(S1 - gred) / np.linalg.norm(gred)
    print("residual distributed / single", r_dist, r_one, "camera step difference", rel(steps[0][:nc], step1[:nc]))
    assert r_dist < 10 * max(r_one, 1e-13)
    assert rel(steps[0][:nc], step1[:nc]) < 1e-8
    # landmark part: each rank back-substitutes its own landmarks
    lay = prob.layout
    full = np.zeros_like(step1)
    full[:nc] = steps[0][:nc]
    for r in range(world):
        cols = (lay.pt_col[owned[r]][:, None] + np.arange(3)[None, :]).ravel()
        full[cols] = steps[r][cols]
    assert rel(full, step1) < 1e-8
    for s in ranks:
        s.close()
    s1.close()


def test_venice_1778_sample_vs_oracle(oracle):
    oracle_sample_check(oracle, "venice-1778", 0.1, "selfcal")


# ---- configs[4]: synthetic-10k, implicit-Schur matrix-free PCG -------------------------------------------------------
def test_synthetic_10k_full_size_implicit_pcg():
    d = pkg.synthetic.make_named("synthetic-10k")
    assert (d.n_cam, d.n_pt) == (10000, 2000000)
    lam = 1e-3
    prob, s = make(d, "selfcal", variant=SchurVariant.Implicit)
    s.with_cg_params(500, 1e-9)   # IterativeSchurSolver::new (implicit_schur.rs:94-95)
    probe_symmetry(s, prob, lam)
    c0 = s.compute_cost()
    step = s.solve_augmented_equation(lam)
    its = s.info()["pcg_iterations"]
    nc = prob.layout.cam_dof
    _, gred = s.get_schur(want_S=False)
    ye, yi = s.schur_matvec(lam, step[:nc])
    r_rel = np.linalg.norm(yi - gred) / max(np.linalg.norm(gred), 1.0)
    print(f"synthetic-10k implicit PCG: {its} iterations, |S dc - g_red| / max(|g_red|, 1) = {r_rel:.2e} "
          f"(explicit tiles agree to {rel(ye, yi):.1e})")
    assert rel(ye, yi) < 1e-10
    # the reference stops on |r| < 1e-9 max(|b|, 1) or after 500 iterations; either way the residual went down a lot
    assert its <= 500 and r_rel < 1e-3
    if its < 500:
        assert r_rel < 1e-8
    # 500 iterations do not reach the reference's 1e-9 on this shape at lambda = 1e-3: the truncated step has a positive
    # predicted reduction (CG minimises the model over its Krylov space) but may raise the true cost, which is what the
    # LM loop is for -- it rejects, raises lambda, and the better-conditioned systems that follow converge.  Five
    # iterations of optimize_with_mode with the implicit variant:
    s.solve_augmented_equation(lam, want_step=False)
    gn, sn, pred = s.step_stats()
    assert pred > 0
    from apex_solver_amd.solver import LevenbergMarquardtConfig
    res, hist, _ = s.lm_optimize(LevenbergMarquardtConfig.for_bundle_adjustment().with_max_iterations(5))
    print("synthetic-10k implicit LM: cost", res.initial_cost, "->", res.final_cost, "accepted", res.successful_steps,
          "rejected", res.unsuccessful_steps, "PCG-bound iterations", hist[:, 1])
    assert res.initial_cost == pytest.approx(c0, rel=1e-12)
    assert res.successful_steps >= 1 and res.final_cost < res.initial_cost
    s.close()


def test_synthetic_10k_sample_vs_oracle(oracle):
    oracle_sample_check(oracle, "synthetic-10k", 0.02, "selfcal", variant=2, cg=(500, 1e-9))
    oracle_sample_check(oracle, "synthetic-10k", 0.02, "selfcal", variant=0)


# ---- the non-banded stress shape: hub cameras + long-range matches, border ordering on the device path ---------------------
@pytest.mark.parametrize("mode", ["selfcal", "ba"])
def test_hub_shape_vs_oracle(oracle, mode):
    """The "-hub" generator at a size the oracle's dense Schur handles (640 cameras, 40 tiles): cameras seen from anywhere
    and accidental long-range matches are ordered last (a dense border of S, 228 tiles instead of 795); the internal
    order is invisible at the boundary -- S, g_red, gradient and the step equal the oracle's with and without it."""
    d = pkg.synthetic.make_problem(640, 16000, 3, 8, config_id=97, window=16, hub_frac=0.04, hub_obs_prob=0.04, long_range_prob=3e-4)
    lam = 1e-3
    out = {}
    for hubs_last in (1, 0):
        prob, s = make(d, mode, opts=(("hubs_last", hubs_last),))
        info = dict(s.info(), border_cameras=s.setup_times()["hub_cameras"])
        step = s.solve_augmented_equation(lam)
        S, gred = s.get_schur()
        out[hubs_last] = (info, step.copy(), S, gred, s.get_gradient())
        s.close()
        print("border", "on :" if hubs_last else "off:", {q: info[q] for q in ("tiles", "etree_levels", "border_cameras")})
    assert out[1][0]["border_cameras"] > 0 and out[0][0]["border_cameras"] == 0
    assert out[1][0]["tiles"] < out[0][0]["tiles"]
    o = oracle.from_data(d, prob.layout, mode=mode, huber_delta=1.0)
    o.linearize()
    ostep, ograd, oS, ogred = o.solve_augmented(lam, 0, want_schur=True)
    nc = prob.layout.cam_dof
    for hubs_last in (1, 0):
        _, step, S, gred, grad = out[hubs_last]
        bwd = np.linalg.norm(oS @ step[:nc] - ogred) / (referee.sym_norm2(oS) * np.linalg.norm(step[:nc]) + np.linalg.norm(ogred))
        errs = dict(S=rel(S, oS), gred=rel(gred, ogred), grad=rel(grad, ograd), step=rel(step, ostep))
        print("hubs_last", hubs_last, {k: f"{v:.1e}" for k, v in errs.items()}, f"backward {bwd:.1e}")
        assert errs["S"] < 1e-12 and errs["gred"] < 1e-10 and errs["grad"] < 1e-12
        assert bwd < 1e-13 and errs["step"] < 1e-7
        if hubs_last == 1:
            exact, qinfo = o.solve_augmented_quad(lam)
            e_gpu, e_64 = rel(step, exact), rel(ostep, exact)
            print(f"referee hub {mode}: |gpu - exact| {e_gpu:.2e}  |fp64 oracle - exact| {e_64:.2e}")
            referee.RECORD.append((f"hub {mode}", e_gpu, e_64))
            assert qinfo["residual"] < 1e-26 and e_gpu <= referee.FP64_ENVELOPE, (e_gpu, e_64)


def test_hub_full_size_properties():
    """final-13682-hub at full size (29 M observations, ~600 border cameras, a 39-tile dense border): the size-independent
    properties -- S dc = g_red through the explicit tiles and through the matrix-free operator (backward error), S
    symmetric positive on probes, one LM iteration reduces the cost -- with the border ordering and the dataflow sweeps
    on a factor of 41 K tiles."""
    d = pkg.synthetic.make_named("final-13682-hub")
    lam = 1e-3
    prob, s = make(d, "selfcal")
    info = dict(s.info(), border_cameras=s.setup_times()["hub_cameras"])
    print("final-13682-hub:", {k: info[k] for k in ("tile_rows", "tiles", "touched_tiles", "etree_levels", "border_cameras")})
    assert info["border_cameras"] > 0
    probe_symmetry(s, prob, lam)
    step = s.solve_augmented_equation(lam)
    bwd_e, bwd_i = camera_step_checks(s, prob, lam, step, "final-13682-hub")
    assert bwd_e < 1e-13 and bwd_i < 1e-12
    s.solve_augmented_equation(lam, want_step=False)
    one_lm_iteration_behaves(s, lam)
    s.close()


# ---- the structure sweep between "banded" and "hubs": a fraction of the landmarks drawn from a global camera popularity ------
def test_mix_shape_vs_oracle(oracle):
    """final-13682-mix:0.05 at 1/20 of the named size (684 cameras; 1/10 until round 5: the dense oracle, the 12 K-square 2-norm and
    the tile-by-tile export of S took 350 s of the GPU suite's 1,200-s limit there -- the structure at 1/10 is now exercised by
    test_refused_plan_selects_the_matrix_free_variant_by_itself): 5 % of the landmarks ignore the capture window and take
    their cameras from a power-law popularity over all cameras (an internet photo collection rather than a capture sequence:
    crates/apex-io/datasets.toml:155-156).  S, g_red, the gradient and the step against the oracle's dense path; the
    border ordering and the fill this structure causes are invisible at the boundary.  Both solvers of the headline JSON:
    the Cholesky variant and the matrix-free fallback (its step against the Cholesky step of the same S)."""
    d = pkg.synthetic.make_named("final-13682-mix:0.05", 0.05)
    lam = 1e-3
    prob, s = make(d, "selfcal")
    info = dict(s.info(), border_cameras=s.setup_times()["hub_cameras"])
    banded = pkg.synthetic.make_named("final-13682", 0.05)
    pb, sb = make(banded, "selfcal")
    ib = sb.info(); sb.close()
    print("mix 0.05:", {k: info[k] for k in ("tile_rows", "tiles", "touched_tiles", "etree_levels", "border_cameras")},
          "| banded:", {k: ib[k] for k in ("tiles", "touched_tiles", "etree_levels")})
    assert info["touched_tiles"] > 2 * ib["touched_tiles"]          # the popular cameras couple far-apart tiles
    o = oracle.from_data(d, prob.layout, mode="selfcal", huber_delta=1.0)
    assert s.compute_cost() == pytest.approx(o.residuals()[0], rel=1e-13)
    o.linearize()
    ostep, ograd, oS, ogred = o.solve_augmented(lam, 0, want_schur=True)
    step = s.solve_augmented_equation(lam)
    S, gred = s.get_schur()
    nc = prob.layout.cam_dof
    bwd = np.linalg.norm(oS @ step[:nc] - ogred) / (referee.sym_norm2(oS) * np.linalg.norm(step[:nc]) + np.linalg.norm(ogred))
    errs = dict(S=rel(S, oS), gred=rel(gred, ogred), grad=rel(s.get_gradient(), ograd), step=rel(step, ostep))
    print("mix 0.05 vs oracle:", {k: f"{v:.1e}" for k, v in errs.items()}, f"backward {bwd:.1e}")
    assert errs["S"] < 1e-12 and errs["gred"] < 1e-10 and errs["grad"] < 1e-12
    assert bwd < 1e-13 and errs["step"] < 1e-6
    one_lm_iteration_behaves(s, lam)
    chol4 = s.solve_augmented_equation(1e4).copy()   # (the device Cholesky at cond(S) <= 1e5: 1e-10 of the oracle's in every parity case)
    s.close()
    # the matrix-free fallback on the same structure, at lambda = 1e4 where its iteration determines the step
    # ... on a handle built MATRIX-FREE ONLY ("matrix_free_only": S never formed, diagonal tiles and no pair list -- what makes
    # the fallback independent of the fill of S) and on an ordinary one: the same iteration, the same step
    steps = {}
    for mfo in (1, 0):
        prob, s = make(d, "selfcal", variant=SchurVariant.Implicit, opts=(("matrix_free_only", mfo),))
        s.with_cg_params(500, 1e-12)
        steps[mfo] = s.solve_augmented_equation(1e4).copy()
        print(f"mix 0.05 matrix-free at lambda 1e4 (matrix_free_only={mfo}):", s.info()["pcg_iterations"], "iterations, tiles", s.info()["tiles"],
              "step vs the Cholesky variant", rel(steps[mfo], chol4))
        assert rel(steps[mfo], chol4) < 1e-8
        if mfo:
            assert s.info()["tiles"] == s.info()["tile_rows"]      # the diagonal tiles, nothing else
            with pytest.raises(pkg.capi.LinAlgError):              # the explicit S does not exist on such a handle
                s.get_schur()
        s.close()
    assert rel(steps[1], steps[0]) < 1e-12


# ---- automatic variant selection: a structure whose direct factorisation is refused must not fail the caller (round 5) ----------
def test_refused_plan_selects_the_matrix_free_variant_by_itself(oracle):
    """final-13682-mix:0.05 through the PLAIN surface (SchurVariant::Sparse, LevenbergMarquardt.optimize) with the plan limit
    lowered so that a small S is refused the way the full-size one is (8e7 tile products: tile_plan.hip): initialize_structure
    succeeds, the handle says which variant it runs and why, and the LM loop converges like the reference's would have
    (its dispatch does not depend on the fill of S: levenberg_marquardt.rs:1039-1082).  At 1/10 of the named size (1,368
    cameras) the structure and the loop; at 1/50 (the oracle's CPU PCG takes 0.35 s per iteration at 1/10) every solve and the
    LM history against the oracle's matrix-free PCG (IterativeSchurSolver at its defaults, implicit_schur.rs:94-95, 835-946).
    With "auto_variant" 0 the refusal is the error it used to be."""
    from apex_solver_amd.solver import LevenbergMarquardt, LevenbergMarquardtConfig

    d = pkg.synthetic.make_named("final-13682-mix:0.05", 0.1)
    with pytest.raises(pkg.capi.LinAlgError, match="tile update list too large"):
        make(d, "selfcal", opts=(("max_tile_updates", 1000), ("auto_variant", 0)))
    prob, s = make(d, "selfcal", opts=(("max_tile_updates", 1000),))
    vi = s.variant_info()
    print("auto variant:", vi, "tiles", s.info()["tiles"], "of", s.info()["tile_rows"], "rows")
    assert vi["variant_asked"] == "Sparse" and vi["variant_used"] == "Implicit" and "refused" in vi["reason"] and "tile update list" in vi["reason"]
    assert s.variant_info(SchurVariant.Iterative)["variant_used"] == "Implicit"
    assert s.info()["tiles"] == s.info()["tile_rows"]          # the diagonal tiles, nothing else: no fill, no pair list
    assert s.setup_times()["pair_slots"] == 0
    s.solve_augmented_equation(1e-3)                             # asked: variant 0
    assert 0 < s.info()["pcg_iterations"] <= 500
    one_lm_iteration_behaves(s, 1e-3)
    with pytest.raises(pkg.capi.LinAlgError):                    # the explicit S does not exist on such a handle
        s.get_schur()
    res = LevenbergMarquardt.with_config(LevenbergMarquardtConfig().with_max_iterations(3)).optimize(prob, solver=s)
    print("auto variant LM at 1/10:", res.status.name, res.iterations, res.initial_cost, "->", res.final_cost)
    assert res.final_cost < 0.5 * res.initial_cost
    s.close()

    # ---- against the oracle, at a size its CPU iteration finishes in seconds --------------------------------------------------
    d = pkg.synthetic.make_named("final-13682-mix:0.05", 0.02)
    prob, s = make(d, "selfcal", opts=(("max_tile_updates", 50),))
    assert s.variant_info()["variant_used"] == "Implicit"
    lam = 1e-3
    o = oracle.from_data(d, prob.layout, mode="selfcal", huber_delta=1.0)
    assert s.compute_cost() == pytest.approx(o.residuals()[0], rel=1e-13)
    o.linearize()
    o.set_cg_params(500, 1e-9)
    istep, ograd = o.solve_augmented(lam, 2)
    it_ora = o.last_pcg_iters
    ostep, _, oS, ogred = o.solve_augmented(lam, 0, want_schur=True)
    step = s.solve_augmented_equation(lam)                       # asked: variant 0
    nc = prob.layout.cam_dof
    it_gpu = s.info()["pcg_iterations"]
    r_gpu = np.linalg.norm(oS @ step[:nc] - ogred); r_ora = np.linalg.norm(oS @ istep[:nc] - ogred)
    print(f"auto variant solve at 1/50: pcg iterations gpu / oracle {it_gpu} / {it_ora}, residual {r_gpu:.2e} / {r_ora:.2e}, "
          f"step vs the oracle's matrix-free step {rel(step, istep):.1e}, vs its Cholesky step {rel(step, ostep):.1e}")
    assert it_gpu > 0 and abs(it_gpu - it_ora) <= max(3, it_ora // 20)
    assert rel(s.get_gradient(), ograd) < 1e-12
    assert r_gpu < 10 * max(r_ora, 1e-9 * max(np.linalg.norm(ogred), 1.0))
    one_lm_iteration_behaves(s, lam)
    # where the iteration determines the step (lambda = 1e4) the fall-back IS the direct solve to the PCG tolerance
    w = s.solve_augmented_equation(1e4)
    wchol, _ = o.solve_augmented(1e4, 0)
    print(f"auto variant at lambda 1e4: {s.info()['pcg_iterations']} iterations, step vs the oracle's Cholesky step {rel(w, wchol):.1e}")
    assert rel(w, wchol) < 1e-7
    s.close()
    # the LM surface with the config a caller of the reference would pass; costs against the oracle's LM with the matrix-free variant
    _, s2 = make(d, "selfcal", opts=(("max_tile_updates", 50),))
    res = LevenbergMarquardt.with_config(LevenbergMarquardtConfig().with_max_iterations(2)).optimize(prob, solver=s2)
    o2 = oracle.from_data(d, prob.layout, mode="selfcal", huber_delta=1.0)
    o2.set_cg_params(500, 1e-9)
    ores = o2.optimize(oracle.LMConfig.default(max_iterations=2, variant=2))
    print("auto variant LM at 1/50:", res.status.name, res.iterations, res.initial_cost, "->", res.final_cost, "| oracle (matrix-free):", ores.status, ores.final_cost)
    assert res.final_cost < 0.5 * res.initial_cost
    assert res.iterations == ores.iterations and res.final_cost == pytest.approx(ores.final_cost, rel=1e-6)
    s2.close()


# ---- variant selection by predicted cost (round 6) ---------------------------------------------------------------------------
def test_variant_is_selected_by_predicted_cost(oracle):
    """Between "cheap" and "refused" the direct factorisation used to be chosen whatever it cost (up to 12.7 s at the plan limit)
    although the handle owns the matrix-free PCG, whose cost does not depend on the fill of S.  apexgpu_set_structure now
    predicts both (apexgpu_variant_costs: the plan's operation counts at the measured rates; the PCG cap times two passes over
    the observations) and builds the cheaper one.  Two structures of the same size on either side of the crossover -- the banded
    final-13682 shape and its mix with 5 % long-range landmarks (S dense at tile granularity), the crossover moved onto this
    small size with "variant_cost_permille" -- through the plain LM surface (SchurVariant::Sparse), each against the oracle's
    matching variant: Cholesky for the one, IterativeSchurSolver's PCG for the other."""
    from apex_solver_amd.solver import LevenbergMarquardt, LevenbergMarquardtConfig

    lo = pkg.synthetic.make_named("final-13682", 0.02)
    hi = pkg.synthetic.make_named("final-13682-mix:0.05", 0.02)
    pred = []
    for d in (lo, hi):
        _, s = make(d, "selfcal")
        vi = s.variant_info()
        assert vi["variant_choice"] == "direct" and vi["predicted_direct_ms"] > 0 and vi["predicted_matrix_free_ms"] > vi["predicted_direct_ms"], vi
        pred.append(vi)
        s.close()
    d_lo, d_hi, mf = pred[0]["predicted_direct_ms"], pred[1]["predicted_direct_ms"], pred[0]["predicted_matrix_free_ms"]
    print(f"predicted ms per solve at 1/50: direct banded {d_lo:.3f}, direct mix {d_hi:.3f}, matrix-free {mf:.3f}")
    assert d_hi > 1.05 * d_lo
    pct = int(np.floor(1000.0 * d_lo / mf)) + 1
    assert d_lo < pct * mf / 1000.0 < d_hi, "no whole per-mille value separates the two structures at this size"
    for d, want, ovar in ((lo, "direct", 0), (hi, "matrix-free by predicted cost", 2)):
        prob, s = make(d, "selfcal", opts=(("variant_cost_permille", pct),))
        vi = s.variant_info()
        print("variant by cost:", d.name, vi)
        assert vi["variant_choice"] == want, vi
        assert vi["variant_used"] == ("Sparse" if ovar == 0 else "Implicit")
        assert ("predicted cost" in vi["reason"]) == (ovar == 2)
        res = LevenbergMarquardt.with_config(LevenbergMarquardtConfig().with_max_iterations(2)).optimize(prob, solver=s)
        o = oracle.from_data(d, prob.layout, mode="selfcal", huber_delta=1.0)
        o.set_cg_params(500, 1e-9)
        ores = o.optimize(oracle.LMConfig.default(max_iterations=2, variant=ovar))
        print("   LM:", res.status.name, res.iterations, res.initial_cost, "->", res.final_cost, "| oracle variant", ovar, ores.final_cost)
        assert res.iterations == ores.iterations and res.final_cost == pytest.approx(ores.final_cost, rel=1e-6)
        s.close()
    # "auto_variant" 0: no plan is refused by cost either
    _, s = make(hi, "selfcal", opts=(("variant_cost_permille", pct), ("auto_variant", 0)))
    assert s.variant_info()["variant_choice"] == "direct"
    s.close()


def test_host_block_cache_lives_with_the_handles():
    """The set-up's big host blocks (observation lists, pair list) are cached for the next set_structure -- but only while a
    handle is alive: apexgpu_destroy of the last one returns everything to the system (apexgpu_host_cache_bytes), and a smaller
    structure after a larger one keeps only what it used (round 6; ADVICE r5: 3.7 GB stayed resident for the life of the process)."""
    import time

    L = pkg.capi.load()

    def wait_for(pred, what):
        for _ in range(200):      # (the blocks come back on the handle's free thread)
            if pred():
                return
            time.sleep(0.05)
        raise AssertionError(what + f": {L.apexgpu_host_cache_bytes()} bytes held")

    L.apexgpu_trim_host_cache(None)
    big = pkg.synthetic.make_named("final-13682", 0.2)      # lists of >= 32 MB: the cached kind
    _, s = make(big, "selfcal")
    wait_for(lambda: L.apexgpu_host_cache_bytes() > 0, "nothing cached behind a set-up")
    held_big = L.apexgpu_host_cache_bytes()
    small = pkg.synthetic.make_named("final-13682", 0.05)
    _, s2 = make(small, "selfcal")
    wait_for(lambda: L.apexgpu_host_cache_bytes() < held_big, "a smaller structure left the larger one's blocks cached")
    s.close()
    assert L.apexgpu_host_cache_bytes() >= 0
    s2.close()
    wait_for(lambda: L.apexgpu_host_cache_bytes() == 0, "the last handle did not release the cache")
