"""Independent numpy/scipy restatement of the SE3 pose-graph path used to cross-check the C oracle when
generating golden fixtures.  Works on rotation MATRICES (scipy Rotation for quaternion <-> matrix and
the rotation logarithm), whereas oracle/pg_oracle.c works on quaternions, so the two share no code path.

Formulas: src/factors/between_factor.rs:268-322 (chain rule), crates/apex-manifolds/src/se3.rs:347-369
(adjoint), :520-558 (Q block as coded), :652-666 (right_jacobian_inv as coded: Jl^-1(theta) on the
diagonal), so3.rs:628-646 (Jl^-1).  TEST INFRASTRUCTURE ONLY.
"""
import numpy as np
from scipy.spatial.transform import Rotation


def hat(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def to_Rt(p):
    q = np.asarray(p[3:7], dtype=np.float64)
    q = q / np.linalg.norm(q)
    return Rotation.from_quat([q[1], q[2], q[3], q[0]]).as_matrix(), np.asarray(p[:3], dtype=np.float64)


def jl_inv(th):
    a = th @ th
    K = hat(th)
    if a <= 1e-10:
        return np.eye(3) - 0.5 * K
    t = np.sqrt(a)
    return np.eye(3) - 0.5 * K + (1.0 / a - (1.0 + np.cos(t)) / (2.0 * t * np.sin(t))) * (K @ K)


def q_block(rho, th):
    Rk, Tk = hat(rho), hat(th)
    t2 = th @ th
    a, b, c, d = 0.5, 1 / 6 + t2 / 120, -1 / 24 + t2 / 720, -1 / 60
    if t2 > 1e-10:
        tn = np.sqrt(t2)
        b = (tn - np.sin(tn)) / tn**3
        c = (1 - t2 / 2 - np.cos(tn)) / tn**4
        d = (c - 3.0) * (tn - np.sin(tn) - tn**3 / 6) / tn**5
    tr, rt = Tk @ Rk, Rk @ Tk
    trt = tr @ Tk
    rtt = rt @ Tk
    return Rk * a + (tr + rt + trt) * b - (rtt - rtt.T - trt * 3.0) * c - (trt @ Tk) * d


def adjoint(R, t):
    A = np.zeros((6, 6))
    A[:3, :3] = R; A[3:, 3:] = R; A[:3, 3:] = hat(t) @ R
    return A


def between(k0, k1, meas):
    R0, t0 = to_Rt(k0); R1, t1 = to_Rt(k1); Rm, tm = to_Rt(meas)
    RA, tA = R1.T @ R0, R1.T @ (t0 - t1)          # k1^-1 k0
    RD, tD = RA @ Rm, RA @ tm + tA                 # (k1^-1 k0) meas
    th = Rotation.from_matrix(RD).as_rotvec()
    D = jl_inv(th)
    rho = D @ tD
    r = np.concatenate([rho, th])
    B = -D @ q_block(-rho, -th) @ D
    j_log = np.block([[D, B], [np.zeros((3, 3)), D]])
    j_diff = adjoint(Rm.T, -Rm.T @ tm)             # Adj(meas^-1)
    j_k1 = -adjoint(RA.T, -RA.T @ tA)              # -Adj((k1^-1 k0)^-1)
    J0 = j_log @ j_diff
    J1 = j_log @ (j_diff @ j_k1)
    return r, np.hstack([J0, J1])


def linearize(poses, e_from, e_to, meas, huber_delta=None):
    n_e = len(e_from)
    r = np.zeros((n_e, 6)); J = np.zeros((n_e, 6, 12))
    for e in range(n_e):
        r[e], J[e] = between(poses[e_from[e]], poses[e_to[e]], meas[e])
        if huber_delta is not None:
            s = r[e] @ r[e]
            if s > huber_delta**2:
                sc = np.sqrt(huber_delta / np.sqrt(s))
                r[e] *= sc; J[e] *= sc
    return r, J


def normal_equations(r, J, e_from, e_to, pose_col):
    n = 6 * len(pose_col)
    H = np.zeros((n, n)); g = np.zeros(n)
    for e in range(len(e_from)):
        cols = np.concatenate([pose_col[e_from[e]] + np.arange(6), pose_col[e_to[e]] + np.arange(6)])
        H[np.ix_(cols, cols)] += J[e].T @ J[e]
        g[cols] += J[e].T @ r[e]
    return H, g
