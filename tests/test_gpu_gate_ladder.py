"""Rows A9 and A12 on the device (`pytest -m gpu`):

* the 3x3 eigenvalue gate of invert_landmark_blocks_with_lambda (explicit_schur.rs:377-442) -- all three regimes, and
  the margins of the trace / determinant shortcut that csrc/ba_device.hpp takes before the eigen-solver -- against
  `ora_invert_landmark_blocks`, both on caller-crafted blocks (apexgpu_debug_invert_blocks runs the kernel's own
  function on the GPU) and on the landmark records a real assembly leaves (apexgpu_get_landmark_blocks);
* the five-step regularisation ladder of solve_with_cholesky (explicit_schur.rs:559-634), forced by an exactly zero
  pivot, against `ora_solve_cholesky`.
"""
import numpy as np
import referee
import pytest

import apex_solver_amd as pkg
from apex_solver_amd import capi
from apex_solver_amd.solver import GpuSchurComplementSolver, OptimizationType, Problem
from apex_solver_amd.synthetic import BAProblemData

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = np.ravel(a); b = np.ravel(b)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def ora_invert(oracle, blocks, lam=0.0):
    b = np.ascontiguousarray(blocks, dtype=np.float64).reshape(-1, 9)
    out = np.empty_like(b)
    rc = oracle.lib().ora_invert_landmark_blocks(len(b), b, float(lam), out)
    return rc, out.reshape(-1, 3, 3)


def sym_from_eigs(rng, e):
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    b = (q * np.asarray(e)) @ q.T
    return 0.5 * (b + b.T)


def regime(b):
    ev = np.linalg.eigvalsh(b)
    if ev[0] < 1e-12:
        return 1
    if ev[2] / ev[0] > 1e10:
        return 2
    return 0


def test_gate_regimes_and_shortcut_margins_on_crafted_blocks(oracle):
    """Blocks with prescribed eigenvalues on both sides of each threshold (min_ev 1e-12, cond 1e10) and of the
    shortcut's own, tighter, trace / determinant bounds (det >= 2e-12 tr^2, tr^3 <= 0.5e10 det)."""
    rng = np.random.default_rng(11)
    blocks = []
    for scale in (1.0, 1e-6, 1e4):
        # regime 1 boundary (an eigenvalue is only known to ~eps max_ev: at max_ev = 1e4 that is 2e-12, the threshold
        # itself, and the reference's own decision is rounding noise there -- the small eigenvalues are probed at <= 1)
        for mn in (0.0, 1e-16, 0.5e-12, 0.9e-12, 1.1e-12, 2e-12, 1e-11, 1e-9) if scale <= 1.0 else (1e-9, 1e-7):
            blocks.append(sym_from_eigs(rng, [mn, 0.3 * scale, scale]))
        for cond in (1e8, 1e9, 0.2e10, 0.45e10, 0.55e10, 0.9e10, 1.1e10, 1e11, 1e13):         # regime 2 boundary
            blocks.append(sym_from_eigs(rng, [scale / cond, 0.5 * scale, scale]))
        for cond in (1e3, 1e5, 1e12, 1e14):                                                  # two small eigenvalues
            blocks.append(sym_from_eigs(rng, [scale / cond, scale / cond * 3, scale]))
        for _ in range(20):                                                                   # ordinary landmarks
            blocks.append(sym_from_eigs(rng, np.sort(rng.uniform(0.1, 10.0, 3) * scale)))
    blocks.append(np.diag([1e-3, 1.0, 2.0]) * 1.0)
    blocks.append(np.zeros((3, 3)))                         # min_ev = 0 -> + 1e-6 I, still invertible
    blocks.append(-np.eye(3) * 1e-6)                        # negative: min_ev < 1e-12 -> reg = 1e-6 + max_ev 1e-6
    # rank-2 J^T J of a single observation (k = 1, lambda = 0): the case regime 1 exists for
    for _ in range(10):
        j = rng.normal(size=(2, 3)) * 300.0
        blocks.append(j.T @ j)
    # near the shortcut's decision lines, where it must hand over to the eigen-solver rather than guess
    for t in np.linspace(0.5, 2.0, 13):
        blocks.append(np.diag([2e-12 * t * 9.0, 1.0, 2.0]))     # det / tr^2 ~ t * 4e-12 ... straddles 2e-12
        blocks.append(np.diag([1.0, 1.0, 1.0]) * 1e-3 + np.diag([0, 0, 1.0]) * (0.5e10 * t) ** (1 / 3.0))
    B = np.stack(blocks)
    got, ok = capi.invert_blocks_on_device(B)
    rc, want = ora_invert(oracle, B)
    assert rc == 0 and ok.all()
    regs = np.array([regime(b) for b in B])
    assert set(regs) == {0, 1, 2}, "the crafted set must reach all three regimes"
    # an independent restatement of the three regimes (numpy eigvalsh + inv) next to the C oracle's
    want_np = []
    for b, rg in zip(B, regs):
        ev = np.linalg.eigvalsh(b)
        reg = {0: 0.0, 1: 1e-6 + ev[2] * 1e-6, 2: ev[2] * 1e-6}[rg]
        want_np.append(np.linalg.inv(b + reg * np.eye(3)))
    # nalgebra's try_inverse is the cofactor formula (restated as such by oracle and device): its rounding error is
    # ~eps max_ev^2 / (min_ev mid_ev) of the matrix actually inverted, and the device contracts FMAs where the host
    # compiler does not, so device and oracle agree to THAT, not to eps.  A wrong regime decision is no rounding-size
    # effect: regularising moves the smallest eigenvalue by a factor >= 1e4, i.e. the inverse changes by O(1).
    worst = {0: 0.0, 1: 0.0, 2: 0.0}
    eps = np.finfo(float).eps
    for b, g, w, rg, wn in zip(B, got, want, regs, want_np):
        ev = np.abs(np.linalg.eigvalsh(np.linalg.inv(wn)))
        ev.sort()
        kappa2 = ev[2] ** 2 / (ev[0] * ev[1])
        tol = min(max(1e-12, 2e4 * eps * kappa2), 1e-2)   # measured: up to ~15x the first-order estimate 200 eps kappa2
        e = rel(g, w)
        worst[rg] = max(worst[rg], e)
        assert e < tol, (rg, e, tol, b)
        assert rel(g, wn) < min(max(1e-4, 100 * tol), 5e-2), (rg, rel(g, wn), tol, b)   # the independent restatement (LAPACK inverse)
    print("3x3 gate on crafted blocks: worst relative difference per regime", worst, "counts", np.bincount(regs))
    # a wrong regime decision is not a rounding-size error: the regularisation changes the inverse by >= 1e-6 relative
    # for every block of regimes 1 and 2 above, so the bounds above cannot hide one


def _custom(n_cam, cam_lists, seed=5, noise=0.7):
    base = pkg.synthetic.make_problem(n_cam, len(cam_lists), 3, 3, config_id=seed)
    cam_idx, pt_idx = [], []
    for l, cams in enumerate(cam_lists):
        cam_idx += list(cams); pt_idx += [l] * len(cams)
    cam_idx = np.asarray(cam_idx, dtype=np.uint32); pt_idx = np.asarray(pt_idx, dtype=np.uint32)
    rng = np.random.default_rng(seed)
    perm = rng.permutation(len(cam_idx))
    cam_idx, pt_idx = cam_idx[perm], pt_idx[perm]
    uv = pkg.synthetic.project_bal(base.truth_poses[cam_idx], base.truth_intr[cam_idx], base.truth_points[pt_idx])
    uv = uv + rng.normal(0, noise, uv.shape)
    return BAProblemData(base.poses, base.intr, base.points, cam_idx, pt_idx, np.ascontiguousarray(uv))


@pytest.mark.parametrize("mode", ["selfcal", "ba"])
@pytest.mark.parametrize("lam", [0.0, 1e-9, 1e-3], ids=["lambda0", "lambda1e-9", "lambda1e-3"])
def test_landmark_records_of_an_assembly_hit_all_regimes(oracle, mode, lam):
    """get_landmark_blocks() of a real assembly against the oracle's inversion of the oracle's own H_ll + lambda I:
    lambda = 0 with k = 1 landmarks is regime 1 (rank-2 block), lambda = 1e-9 with k = 1 is regime 2 (cond ~ 1e13),
    k >= 3 well-spread landmarks are regime 0."""
    n_cam = 24
    rng = np.random.default_rng(3)
    lists = [[c] for c in range(8)]                                       # k = 1
    lists += [[c, (c + 1) % n_cam] for c in range(8)]                     # k = 2, neighbouring cameras
    lists += [sorted(rng.choice(n_cam, size=int(k), replace=False).tolist()) for k in rng.integers(3, 9, size=200)]
    d = _custom(n_cam, lists)
    ot = OptimizationType.SelfCalibration if mode == "selfcal" else OptimizationType.BundleAdjustment
    prob = Problem.bundle_adjustment(d, ot, 1.0)
    s = GpuSchurComplementSolver(0).initialize_structure(prob)
    s.set_parameters(d.poses, d.intr, d.points)
    s.assemble(lam)
    hinv, gl = s.get_landmark_blocks()
    o = oracle.from_data(d, prob.layout, mode=mode, huber_delta=1.0)
    _, r, Jp, Jl, Ji = o.linearize()
    H = np.zeros((d.n_pt, 3, 3)); g = np.zeros((d.n_pt, 3))
    np.add.at(H, d.pt_idx, np.einsum("nra,nrb->nab", Jl, Jl))
    np.add.at(g, d.pt_idx, np.einsum("nra,nr->na", Jl, r.reshape(-1, 2)))
    H += lam * np.eye(3)                                                  # explicit_schur.rs:1205-1212
    rc, want = ora_invert(oracle, H)                                      # called with lambda = 0.0 (:365-367, 1215)
    assert rc == 0
    regs = np.array([regime(b) for b in H])
    print(mode, "lambda", lam, "regimes", np.bincount(regs, minlength=3))
    # k = 1, lambda = 0: the rank-2 block's smallest eigenvalue is rounding noise of size eps max_ev ~ 1e-11 around 0, so
    # the reference's own decision between regimes 1 and 2 is noise; either way the block is regularised
    if lam == 0.0:
        assert np.isin(regs[:8], (1, 2)).all()
    if lam == 1e-9:
        assert (regs[:8] == 2).all()
    assert (regs[16:] == 0).sum() > 150
    assert rel(gl, g) < 1e-12
    eps = np.finfo(float).eps
    worst = 0.0
    for l in range(d.n_pt):
        ev = np.abs(np.linalg.eigvalsh(np.linalg.inv(want[l]))); ev.sort()
        tol = min(max(1e-11, 2e4 * eps * ev[2] ** 2 / (ev[0] * ev[1])), 1e-2)   # cofactor formula, see the crafted-block test
        e = rel(hinv[l], want[l])
        if lam == 0.0 and l < 8 and e >= tol:     # the other noise-driven regime: reg differs by the constant 1e-6
            mx = np.linalg.eigvalsh(H[l])[2]
            alt = [np.linalg.inv(H[l] + r * np.eye(3)) for r in (1e-6 + mx * 1e-6, mx * 1e-6)]
            e = min(rel(hinv[l], a) for a in alt)
        worst = max(worst, e)
        assert e < tol, (l, regs[l], e, tol)
    print("landmark records vs oracle: worst relative difference", worst)
    s.close()


@pytest.mark.parametrize("mode", ["selfcal", "ba"])
def test_cholesky_ladder_on_a_zero_pivot(oracle, mode):
    """A camera that no factor touches has a zero diagonal block in S when lambda = 0: the first Cholesky meets an
    exactly zero pivot, the ladder (explicit_schur.rs:559-634) re-solves S + reg I with
    reg = max(trace / n, max |diag|, 1) 10^(k-4).  Same attempt, same reg and the same step as the oracle."""
    n_cam = 40
    rng = np.random.default_rng(17)
    cams = [c for c in range(n_cam) if c != 23]                          # camera 23 observes nothing
    lists = [sorted(rng.choice(cams, size=int(k), replace=False).tolist()) for k in rng.integers(3, 8, size=900)]
    d = _custom(n_cam, lists, seed=7)
    ot = OptimizationType.SelfCalibration if mode == "selfcal" else OptimizationType.BundleAdjustment
    prob = Problem.bundle_adjustment(d, ot, 1.0)
    s = GpuSchurComplementSolver(0).initialize_structure(prob)
    s.set_parameters(d.poses, d.intr, d.points)
    o = oracle.from_data(d, prob.layout, mode=mode, huber_delta=1.0)
    o.linearize()
    lam = 0.0
    ostep, ograd, oS, ogred = o.solve_augmented(lam, 0, want_schur=True)
    assert o.last_reg > 0.0, "the oracle must have climbed the ladder too"
    step = s.solve_augmented_equation(lam)
    reg = s.info()["last_reg"]
    print(mode, "ladder: reg gpu / oracle", reg, o.last_reg)
    assert reg == pytest.approx(o.last_reg, rel=1e-12)
    nc = prob.layout.cam_dof
    A = oS + o.last_reg * np.eye(nc)
    bwd = np.linalg.norm(A @ step[:nc] - ogred) / (referee.sym_norm2(A) * np.linalg.norm(step[:nc]) + np.linalg.norm(ogred))
    err = rel(step, ostep)
    print(mode, "ladder: backward error", bwd, "step vs oracle", err, "cond(S + reg I)", np.linalg.cond(A))
    assert bwd < 1e-13 and err < 1e-10
    assert rel(s.get_gradient(), ograd) < 1e-12
    # the step statistics and the trial cost of the ladder's step (the speculative path enqueued them behind a failed factor:
    # they must be those of the repeated solve)
    gn, sn, pred = s.step_stats()
    trial = s.eval_step()
    s.discard_step()
    # and with damping the same problem needs no ladder
    s.solve_augmented_equation(1e-3)
    assert s.info()["last_reg"] == 0.0
    s.close()
    # the same failed pivot with three host waits per solve ("one_wait" 0: the flags are read before anything is enqueued behind
    # the factorisation) and without the eager step evaluation: same ladder, same step, same statistics, same trial cost
    for opts in ({"one_wait": 0}, {"eager_step_eval": 0}, {"one_wait": 0, "eager_step_eval": 0}):
        s2 = GpuSchurComplementSolver(0)
        for k, v in opts.items():
            s2.with_option(k, v)
        s2.initialize_structure(prob)
        s2.set_parameters(d.poses, d.intr, d.points)
        step2 = s2.solve_augmented_equation(lam)
        assert s2.info()["last_reg"] == pytest.approx(reg, rel=1e-15), opts
        assert np.array_equal(step2, step), (opts, rel(step2, step))
        gn2, sn2, pred2 = s2.step_stats()
        assert (gn2, sn2, pred2) == pytest.approx((gn, sn, pred), rel=1e-13), opts
        assert s2.eval_step() == pytest.approx(trial, rel=1e-13), opts
        s2.discard_step()
        s2.close()
