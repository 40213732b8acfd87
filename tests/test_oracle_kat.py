"""Pin the CPU oracle against every known-answer unit test the reference holds for the
bundle-adjustment hot path.  Each test names the reference test it
transcribes (file:line under /root/reference); numbers and tolerances are the reference's.
"""
import ctypes as C

import numpy as np
import pytest


def _inv(ora, blocks, lam=0.0):
    L = ora.lib()
    b = np.ascontiguousarray(np.asarray(blocks, dtype=np.float64).reshape(-1, 9))
    out = np.empty_like(b)
    rc = L.ora_invert_landmark_blocks(b.shape[0], b, lam, out)
    return rc, out.reshape(-1, 3, 3)


def _rows(dense_hcl):
    """The reference's merged row lists (explicit_schur.rs:811-865) from a dense H_cl."""
    n_c, ncol = dense_hcl.shape
    n_pt = ncol // 3
    ptr, rows, vals = [0], [], []
    for l in range(n_pt):
        blk = dense_hcl[:, 3 * l:3 * l + 3]
        for i in range(n_c):
            if np.any(blk[i] != 0):
                rows.append(i)
                vals.append(blk[i])
        ptr.append(len(rows))
    return (np.asarray(ptr, dtype=np.int64), np.asarray(rows + [0], dtype=np.int64),
            np.ascontiguousarray(np.asarray(vals + [[0, 0, 0]], dtype=np.float64)))


# --- explicit_schur.rs ------------------------------------------------------------
def test_3x3_block_inversion(oracle):
    """explicit_schur.rs:1454-1459"""
    rc, inv = _inv(oracle, [np.diag([2.0, 3.0, 4.0])])
    assert rc == 0 and abs(inv[0][0, 0] - 0.5) < 1e-10


def test_invert_landmark_blocks_with_lambda(oracle):
    """explicit_schur.rs:1647-1663"""
    rc, inv = _inv(oracle, [np.diag([2.0, 3.0, 4.0])], 0.0)
    assert rc == 0
    assert abs(inv[0][0, 0] - 0.5) < 1e-10 and abs(inv[0][1, 1] - 1 / 3) < 1e-10 and abs(inv[0][2, 2] - 0.25) < 1e-10
    rc, inv = _inv(oracle, [np.diag([2.0, 3.0, 4.0])], 1.0)
    assert abs(inv[0][0, 0] - 0.5) < 1e-10


def test_invert_gate_regimes(oracle):
    """The three regimes of explicit_schur.rs:399-427 (thresholds 1e-12, cond 1e10, scale 1e-6)."""
    # singular (min_ev < 1e-12): inv(B + (max(0,1e-6) + max_ev*1e-6) I)
    B = np.diag([4.0, 1.0, 0.0])
    rc, inv = _inv(oracle, [B])
    reg = 1e-6 + 4.0 * 1e-6
    assert rc == 0 and np.allclose(inv[0], np.linalg.inv(B + reg * np.eye(3)), rtol=1e-12)
    # ill-conditioned (cond > 1e10, min_ev >= 1e-12): inv(B + max_ev*1e-6 I)
    B = np.diag([1e3, 1.0, 1e-9])
    rc, inv = _inv(oracle, [B])
    assert rc == 0 and np.allclose(inv[0], np.linalg.inv(B + 1e3 * 1e-6 * np.eye(3)), rtol=1e-12)
    # well conditioned, non-diagonal: plain inverse
    rng = np.random.default_rng(1)
    A = rng.standard_normal((3, 3)); B = A @ A.T + np.eye(3)
    rc, inv = _inv(oracle, [B])
    assert rc == 0 and np.allclose(inv[0], np.linalg.inv(B), rtol=1e-12)


def test_compute_schur_complement_known_matrix(oracle):
    """explicit_schur.rs:1473-1515: S(0,0)=3.5, S(1,1)=3.0"""
    L = oracle.lib()
    Hcc = np.diag([4.0, 5.0])
    Hcl = np.zeros((2, 3)); Hcl[0, 0] = 1.0; Hcl[1, 1] = 2.0
    ptr, rows, vals = _rows(Hcl)
    S = np.empty((2, 2))
    L.ora_schur_complement(2, np.ascontiguousarray(Hcc), 1, ptr, rows, vals, np.ascontiguousarray(0.5 * np.eye(3).ravel()), S)
    assert abs(S[0, 0] - 3.5) < 1e-10 and abs(S[1, 1] - 3.0) < 1e-10


def test_back_substitute(oracle):
    """explicit_schur.rs:1518-1546: delta_p = [0,0,3]"""
    L = oracle.lib()
    Hcl = np.zeros((2, 3)); Hcl[0, 0] = 1.0; Hcl[1, 1] = 1.0
    ptr, rows, vals = _rows(Hcl)
    dp = np.empty(3)
    L.ora_back_substitute(1, np.array([1.0, 2.0]), np.array([1.0, 2.0, 3.0]), ptr, rows, vals, np.eye(3).ravel().copy(), dp)
    assert abs(dp[0]) < 1e-10 and abs(dp[1]) < 1e-10 and abs(dp[2] - 3.0) < 1e-10


def test_compute_reduced_gradient(oracle):
    """explicit_schur.rs:1549-1576: g_reduced = [-1,-2]"""
    L = oracle.lib()
    Hcl = np.zeros((2, 3)); Hcl[0, 0] = 1.0; Hcl[1, 1] = 1.0
    ptr, rows, vals = _rows(Hcl)
    g = np.empty(2)
    L.ora_reduced_gradient(2, np.array([1.0, 2.0]), 1, np.array([1.0, 2.0, 3.0]), ptr, rows, vals,
                           (2 * np.eye(3)).ravel().copy(), g)
    assert abs(g[0] + 1.0) < 1e-10 and abs(g[1] + 2.0) < 1e-10


def test_solve_with_cholesky_small_spd(oracle):
    """explicit_schur.rs:1914-1938: [[4,1],[1,3]] x = [1,2], 1e-8"""
    L = oracle.lib()
    A = np.array([[4.0, 1.0], [1.0, 3.0]]); b = np.array([1.0, 2.0]); x = np.empty(2)
    assert L.ora_solve_cholesky(2, A, b, x, None) == 0
    assert np.all(np.abs(A @ x - b) < 1e-8)


def test_solve_with_cholesky_regularisation_ladder(oracle):
    """explicit_schur.rs:559-634: indefinite S -> retry with base*10^(k-4)."""
    L = oracle.lib()
    b = np.array([1.0, 1.0]); x = np.empty(2); reg = C.c_double(0)
    # base = max(trace/n, max|diag|, 1) = 1 -> levels 1e-4, 1e-3, 1e-2, 1e-1, 1
    A2 = np.array([[1.0, 1.2], [1.2, 1.0]])  # eigenvalues 2.2, -0.2 -> reg=1.0 works (1e-1 does not)
    assert L.ora_solve_cholesky(2, A2, b, x, C.addressof(reg)) == 0
    assert reg.value == pytest.approx(1.0)
    assert np.allclose((A2 + np.eye(2)) @ x, b)


def test_solve_with_cholesky_exhausted(oracle):
    L = oracle.lib()
    A = np.array([[1.0, 3.0], [3.0, 1.0]])  # eigenvalues 4, -2: even reg = 1.0 is not enough
    x = np.empty(2)
    assert L.ora_solve_cholesky(2, A, np.array([1.0, 1.0]), x, None) == -2  # SingularMatrix


def test_solve_with_pcg_diagonal_system(oracle):
    """explicit_schur.rs:1942-1959: diag(2,3) x = [1,2] -> [1/2, 2/3], 1e-6"""
    L = oracle.lib()
    x = np.empty(2); it = C.c_int64(0)
    L.ora_solve_pcg(2, np.diag([2.0, 3.0]), np.array([1.0, 2.0]), 200, 1e-6, x, C.byref(it))
    assert abs(x[0] - 0.5) < 1e-6 and abs(x[1] - 2 / 3) < 1e-6


def _schur_fixture():
    """create_schur_test_setup (explicit_schur.rs:1304-1363): 36x21 0/1 Jacobian."""
    J = np.zeros((36, 21))
    for ci, cam_col in enumerate([0, 6]):
        for li, lm_col in enumerate([12, 15, 18]):
            rb = (ci * 3 + li) * 6
            for k in range(6):
                J[rb + k, cam_col + k] = 1.0
                J[rb + k, lm_col + (k % 3)] = 1.0
    r = np.array([(i % 5) * 0.1 for i in range(36)])
    return J, r


def _fixture_solve(ora, lam, variant=0):
    J, r = _schur_fixture()
    step = np.zeros(21); grad = np.zeros(21)
    rc = ora.lib().ora_schur_solve_dense_jacobian(36, 12, 3, np.ascontiguousarray(J), r, lam, variant, 200, 1e-6, step, grad)
    return rc, step, grad, J, r


def test_schur_fixture_structure(oracle):
    """The fixture guarantees H_cc = 3 I12 and H_pp = 4 I3 (explicit_schur.rs:1303)."""
    J, _ = _schur_fixture()
    H = J.T @ J
    assert np.allclose(H[:12, :12], 3 * np.eye(12)) and np.allclose(H[12:, 12:], 4 * np.eye(9))


def test_iterative_schur_fixture(oracle):
    """implicit_schur.rs:1231-1260, 1371-1394 on the same fixture: 21-vector, different lambda -> different
    update; and, stronger, the matrix-free PCG lands on the damped normal equations' solution."""
    sols = []
    for lam in (0.001, 100.0):
        rc, step, grad, J, r = _fixture_solve(oracle, lam, 2)
        assert rc == 0 and step.shape == (21,)
        H = J.T @ J + lam * np.eye(21)
        assert np.linalg.norm(H @ step + J.T @ r) <= 1e-5 * max(np.linalg.norm(J.T @ r), 1.0)
        sols.append(step)
    assert np.sum((sols[0] - sols[1]) ** 2) > 1e-10


@pytest.mark.parametrize("variant", [0, 1])
def test_explicit_schur_augmented_solves_damped_normal_equations(oracle, variant):
    """solve_augmented_equation (explicit_schur.rs:1129-1234) == (J^T J + lambda I) dx = -J^T r."""
    for lam in (1e-3, 100.0):
        rc, step, grad, J, r = _fixture_solve(oracle, lam, variant)
        assert rc == 0
        ref = np.linalg.solve(J.T @ J + lam * np.eye(21), -J.T @ r)
        assert np.allclose(grad, J.T @ r, atol=1e-14)  # get_gradient() is +J^T r
        assert np.allclose(step, ref, rtol=0, atol=1e-5 if variant else 1e-10)


def test_explicit_schur_augmented_lambda_effect(oracle):
    """explicit_schur.rs:1767-1797"""
    _, d1, _, _, _ = _fixture_solve(oracle, 0.001)
    _, d2, _, _, _ = _fixture_solve(oracle, 100.0)
    assert np.sum((d1 - d2) ** 2) > 1e-10


# --- camera: bal_pinhole.rs ----------------------------------------------------------
def test_bal_strict_projection_at_optical_axis(oracle):
    """bal_pinhole.rs:818-829"""
    uv = np.empty(2)
    assert oracle.lib().ora_bal_project(np.array([500.0, 0, 0]), np.array([0.0, 0.0, -1.0]), uv) == 1
    assert abs(uv[0]) < 1e-10 and abs(uv[1]) < 1e-10


def test_bal_strict_projection_off_axis(oracle):
    """bal_pinhole.rs:831-844: (0.1,0.2,-1), f=500 -> (50,100)"""
    uv = np.empty(2)
    assert oracle.lib().ora_bal_project(np.array([500.0, 0, 0]), np.array([0.1, 0.2, -1.0]), uv) == 1
    assert abs(uv[0] - 50.0) < 1e-10 and abs(uv[1] - 100.0) < 1e-10


def test_project_returns_error_behind_camera(oracle):
    """bal_pinhole.rs:957-962 and the MIN_DEPTH boundary (lib.rs:80)."""
    L = oracle.lib(); uv = np.empty(2); i = np.array([500.0, 0, 0])
    assert L.ora_bal_project(i, np.array([0.0, 0.0, 1.0]), uv) == 0
    assert L.ora_bal_project(i, np.array([0.0, 0.0, -1e-6]), uv) == 0  # z < -1e-6 is strict
    assert L.ora_bal_project(i, np.array([0.0, 0.0, -1.1e-6]), uv) == 1


def test_jacobian_pose_numerical(oracle):
    """bal_pinhole.rs:904-954: central differences eps 1e-7, tol 1e-5 on |a-n|/(1+|n|),
    right perturbation pose' = pose * Exp(delta) (constants lib.rs:62,68)."""
    L = oracle.lib()
    pose = np.array([0.1, -0.05, 0.2, 1.0, 0.0, 0.0, 0.0])  # from_translation_euler(..., 0,0,0)
    intr = np.array([500.0, 0.0, 0.0])
    pw = np.array([0.1, 0.05, -3.0])
    r = np.empty(2); Jp = np.empty(12); Jl = np.empty(6); Ji = np.empty(6)
    assert L.ora_linearize_obs(pose, intr, pw, np.zeros(2), -1.0, 1, r, Jp, Jl, Ji) == 1
    Jp = Jp.reshape(2, 6)
    eps = 1e-7

    def uv_at(delta):
        out = np.empty(7)
        L.ora_se3_plus(pose, np.ascontiguousarray(delta), out)
        rr = np.empty(2)
        L.ora_linearize_obs(out, intr, pw, np.zeros(2), -1.0, 0, rr, None, None, None)
        return rr.copy()

    for i in range(6):
        d = np.zeros(6); d[i] = eps
        num = (uv_at(d) - uv_at(-d)) / (2 * eps)
        assert np.all(np.abs(Jp[:, i] - num) / (1 + np.abs(num)) < 1e-5)


def test_jacobian_point_and_intrinsics_numerical(oracle):
    L = oracle.lib()
    pose = np.array([0.1, -0.05, 0.2, 0.9, 0.1, -0.2, 0.3])
    intr = np.array([700.0, 0.01, -0.002])
    pw = np.array([0.3, -0.25, -4.0])
    r = np.empty(2); Jp = np.empty(12); Jl = np.empty(6); Ji = np.empty(6)
    assert L.ora_linearize_obs(pose, intr, pw, np.zeros(2), -1.0, 1, r, Jp, Jl, Ji) == 1

    def uv(pw_, intr_):
        rr = np.empty(2)
        L.ora_linearize_obs(pose, np.ascontiguousarray(intr_), np.ascontiguousarray(pw_), np.zeros(2), -1.0, 0, rr, None, None, None)
        return rr.copy()

    eps = 1e-7
    for i in range(3):
        d = np.zeros(3); d[i] = eps
        num = (uv(pw + d, intr) - uv(pw - d, intr)) / (2 * eps)
        assert np.all(np.abs(Jl.reshape(2, 3)[:, i] - num) / (1 + np.abs(num)) < 1e-5)
        h = eps * max(1.0, abs(intr[i]))
        di = np.zeros(3); di[i] = h
        num = (uv(pw, intr + di) - uv(pw, intr - di)) / (2 * h)
        assert np.all(np.abs(Ji.reshape(2, 3)[:, i] - num) / (1 + np.abs(num)) < 1e-5)


# --- factor: projection_factor.rs:396-522 ------------------------------------------------
def test_projection_factor_zero_residual_at_true_projection(oracle):
    L = oracle.lib()
    pose = np.array([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0]); intr = np.array([500.0, 0.0, 0.0])
    pw = np.array([0.1, 0.2, -1.0]); uv = np.empty(2)
    L.ora_bal_project(intr, pw, uv)
    r = np.empty(2); Jp = np.empty(12); Jl = np.empty(6); Ji = np.empty(6)
    assert L.ora_linearize_obs(pose, intr, pw, uv, 1.0, 1, r, Jp, Jl, Ji) == 1
    assert np.all(np.abs(r) < 1e-10)


def test_projection_factor_behind_camera_zero_residual_and_jacobian(oracle):
    """projection_factor.rs:227-238: invalid projection -> r = 0, Jacobian rows 0."""
    L = oracle.lib()
    pose = np.array([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0]); intr = np.array([500.0, 0.0, 0.0])
    r = np.ones(2); Jp = np.ones(12); Jl = np.ones(6); Ji = np.ones(6)
    assert L.ora_linearize_obs(pose, intr, np.array([0.1, 0.2, 1.0]), np.array([3.0, 4.0]), 1.0, 1, r, Jp, Jl, Ji) == 0
    assert not r.any() and not Jp.any() and not Jl.any() and not Ji.any()


# --- corrector.rs:309-349, loss_functions.rs:364-380 -------------------------------------
def test_corrector_huber_inlier(oracle):
    rs = C.c_double(0); a2 = C.c_double(0)
    sq = oracle.lib().ora_huber_corrector(1.0, 0.06, C.byref(rs), C.byref(a2))
    assert abs(sq - 1.0) < 1e-10 and abs(a2.value) < 1e-10 and abs(rs.value - 1.0) < 1e-10


def test_corrector_huber_outlier(oracle):
    rs = C.c_double(0); a2 = C.c_double(0)
    sq = oracle.lib().ora_huber_corrector(1.0, 75.0, C.byref(rs), C.byref(a2))
    assert 0.0 < sq < 1.0 and a2.value == 0.0
    assert sq == pytest.approx(np.sqrt(1.0 / np.sqrt(75.0)), rel=1e-15)
    assert rs.value == sq  # rho'' < 0 -> residual_scaling = sqrt(rho') (corrector.rs:156-162)


# --- optimizer ------------------------------------------------------------------------
def test_compute_cost_known_value(oracle):
    """optimizer/mod.rs:991-997: cost([1,2]) = 2.5"""
    assert abs(oracle.lib().ora_compute_cost(2, np.array([1.0, 2.0])) - 2.5) < 1e-12


def test_update_damping_accepted_step(oracle):
    """levenberg_marquardt.rs:1566-1590"""
    lam = C.c_double(1e-2); nu = C.c_double(8.0)
    assert oracle.lib().ora_update_damping(0.8, C.byref(lam), C.byref(nu), 1e-15, 1e15) == 1
    assert lam.value < 1e-2 and abs(nu.value - 2.0) < 1e-15
    assert lam.value == pytest.approx(1e-2 * max(1 / 3, 1 - (2 * 0.8 - 1) ** 3), rel=1e-15)


def test_update_damping_rejected_step(oracle):
    """levenberg_marquardt.rs:1592-1618"""
    lam = C.c_double(1e-2); nu = C.c_double(2.0)
    assert oracle.lib().ora_update_damping(-0.5, C.byref(lam), C.byref(nu), 1e-15, 1e15) == 0
    assert lam.value == pytest.approx(2e-2) and abs(nu.value - 4.0) < 1e-15


def test_step_quality(oracle):
    """optimizer/mod.rs:668-675"""
    L = oracle.lib()
    assert L.ora_step_quality(10.0, 8.0, 4.0) == pytest.approx(0.5)
    assert L.ora_step_quality(10.0, 8.0, 1e-16) == 1.0
    assert L.ora_step_quality(10.0, 12.0, 1e-16) == 0.0


# --- manifold: se3.rs:1111-1134 -----------------------------------------------------------
def test_se3_specific_values(oracle):
    L = oracle.lib()
    r = np.empty(2)
    # translation only: act(0) = t ; check through the projection of a point at the origin
    pose = np.array([1.0, 2.0, -3.0, 1.0, 0.0, 0.0, 0.0])
    L.ora_linearize_obs(pose, np.array([1.0, 0.0, 0.0]), np.zeros(3), np.zeros(2), -1.0, 0, r, None, None, None)
    assert np.allclose(r, [1.0 / 3.0, 2.0 / 3.0], atol=1e-12)
    # 90 deg roll maps (0,1,0) -> (0,0,1): from_euler_angles(pi/2,0,0) = (cos45, sin45, 0, 0)
    c = np.sqrt(0.5)
    pose = np.array([0.0, 0.0, -2.0, c, c, 0.0, 0.0])
    L.ora_linearize_obs(pose, np.array([1.0, 0.0, 0.0]), np.array([0.0, 1.0, 0.0]), np.zeros(2), -1.0, 0, r, None, None, None)
    # p_cam = (0,0,1) + (0,0,-2) = (0,0,-1) -> uv = (0,0)
    assert np.allclose(r, [0.0, 0.0], atol=1e-12)


def test_se3_plus_small_and_large_angle_agree_with_rodrigues(oracle):
    """so3.rs:558-612 both branches (theta^2 <= 1e-10 and >) against scipy."""
    from scipy.spatial.transform import Rotation as Rot
    import np_ref

    L = oracle.lib()
    pose = np.array([0.3, -0.2, 0.5, 0.9, 0.1, -0.2, 0.3]); pose[3:] /= np.linalg.norm(pose[3:])
    for scale in (1e-6, 1e-2, 0.7):
        d = scale * np.array([0.3, -0.5, 0.2, 0.4, 0.1, -0.3])
        out = np.empty(7)
        L.ora_se3_plus(pose, d, out)
        R = np_ref.quat_to_R(pose[None, 3:])[0]
        Rn = R @ Rot.from_rotvec(d[3:]).as_matrix()
        tn = pose[:3] + R @ (np_ref.so3_V(d[None, 3:])[0] @ d[:3])
        assert np.allclose(np_ref.quat_to_R(out[None, 3:])[0], Rn, atol=1e-12)
        assert np.allclose(out[:3], tn, atol=1e-12)


# --- the sparse form of solve_with_cholesky (the timed CPU baseline) against the dense one ---------
def test_sparse_cholesky_agrees_with_dense(oracle):
    """explicit_schur.rs:913-921 (entries |v| <= 1e-12 dropped) + :544-550 (sparse LL^T in a fill-reducing order):
    ora_solve_cholesky_sparse restates that contract with reverse Cuthill-McKee + an envelope Cholesky; parity uses the
    dense ora_solve_cholesky.  The two differ by the order of the sums only -- on a block-banded system with a wrapped
    corner (the ring shape of the synthetic BAL generators), an arrow (hub) system, and a dense one; the ordering must
    find the band, and a matrix that needs the regularisation ladder must walk it to the same rung."""
    L = oracle.lib()
    rng = np.random.default_rng(7)

    def solve_both(S, blk):
        n = S.shape[0]
        b = rng.standard_normal(n)
        x0 = np.empty(n); x1 = np.empty(n)
        r0 = C.c_double(0.0); r1 = C.c_double(0.0); st = (C.c_double * 3)()
        assert L.ora_solve_cholesky(n, S, b, x0, C.byref(r0)) == 0
        assert L.ora_solve_cholesky_sparse(n, blk, S, b, x1, C.byref(r1), C.byref(st)) == 0
        return x0, x1, r0.value, r1.value, [st[0], st[1], st[2]], b

    # (a) ring of 40 blocks of 6, half bandwidth 3 blocks, wrapped
    nb, blk = 40, 6
    n = nb * blk
    A = np.zeros((n, n))
    for i in range(nb):
        for d in range(-3, 4):
            j = (i + d) % nb
            A[i * blk:(i + 1) * blk, j * blk:(j + 1) * blk] = rng.standard_normal((blk, blk))
    S = A @ A.T + 50.0 * np.eye(n)
    S[np.abs(S) < 1e-9] = 0.0
    x0, x1, r0, r1, st, b = solve_both(np.ascontiguousarray(S), blk)
    assert r0 == 0.0 and r1 == 0.0
    assert np.linalg.norm(x1 - x0) <= 1e-12 * np.linalg.norm(x0)
    assert np.linalg.norm(S @ x1 - b) <= 1e-12 * np.linalg.norm(b)
    assert st[2] <= 2 * 2 * 6 * blk + blk, st      # RCM on a ring: at most twice the ring's own band (here 6 blocks each side)
    assert st[1] < 0.45 * n * n                     # ... so the envelope is a fraction of the dense triangle pair

    # (b) arrow: 30 independent blocks + one hub block coupled to all of them
    nb, blk = 31, 3
    n = nb * blk
    S = np.zeros((n, n))
    for i in range(nb):
        M = rng.standard_normal((blk, blk)); S[i * blk:(i + 1) * blk, i * blk:(i + 1) * blk] = M @ M.T + 4.0 * np.eye(blk)
    for i in range(nb - 1):
        Cb = 0.3 * rng.standard_normal((blk, blk))
        S[(nb - 1) * blk:, i * blk:(i + 1) * blk] = Cb; S[i * blk:(i + 1) * blk, (nb - 1) * blk:] = Cb.T
    S[(nb - 1) * blk:, (nb - 1) * blk:] += 30.0 * np.eye(blk)
    x0, x1, r0, r1, st, b = solve_both(np.ascontiguousarray(S), blk)
    assert np.linalg.norm(x1 - x0) <= 1e-12 * np.linalg.norm(x0) and st[0] == 2 * nb - 1

    # (c) dense, and a block size that does not divide n (falls back to scalar granularity)
    M = rng.standard_normal((37, 37)); S = np.ascontiguousarray(M @ M.T + 37.0 * np.eye(37))
    x0, x1, r0, r1, st, b = solve_both(S, 5)
    assert np.linalg.norm(x1 - x0) <= 1e-12 * np.linalg.norm(x0) and st[1] == 37 * 38 / 2

    # (d) the regularisation ladder (:559-634): a rank-deficient S takes the same rung in both
    v = rng.standard_normal((24, 5)); S = np.ascontiguousarray(v @ v.T)
    x0, x1, r0, r1, st, b = solve_both(S, 3)
    assert r0 > 0.0 and r1 == r0
    assert np.linalg.norm(x1 - x0) <= 1e-6 * np.linalg.norm(x0)


def test_sparse_cholesky_step_on_a_ba_problem(oracle):
    """variant 3 of ora_solve_augmented (what bench.py's cpu_baseline times) against variant 0 on a synthetic BA problem:
    the same step to the conditioning of S, and an envelope well below the dense triangle on the ring shape."""
    import apex_solver_amd as pkg

    d = pkg.synthetic.make_problem(150, 3000, 3, 7, config_id=11, window=12)
    lay = pkg.layout.reference_column_layout(d.n_cam, d.n_pt)
    o = oracle.from_data(d, lay, mode="selfcal")
    o.residuals(); o.linearize()
    for lam, tol in ((1e4, 1e-12), (1e-3, 1e-7)):
        s0, g0 = o.solve_augmented(lam, 0)
        s3, g3 = o.solve_augmented(lam, 3)
        assert np.array_equal(g0, g3)
        assert np.linalg.norm(s3 - s0) <= tol * np.linalg.norm(s0), (lam, np.linalg.norm(s3 - s0) / np.linalg.norm(s0))
    st = o.last_sparse_stats()
    n = o.cam_dof
    assert st["envelope_entries"] < 0.5 * n * (n + 1) / 2 and st["half_bandwidth"] < n // 2, st
