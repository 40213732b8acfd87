"""Independent numpy/scipy restatement of the reference BA iteration.

Purpose: pin the C oracle (oracle/ba_oracle.c).  It is written against the
reference's *mathematical* definition with different formulations from the oracle
(rotation matrices instead of quaternion sandwiches, a sparse J in the reference's
global column order, H = J^T J by scipy.sparse, and a DIRECT solve of the full damped
normal equations (H + lambda I) dx = -J^T r instead of the Schur elimination), so an
agreement between the two is evidence about the algorithm, not about shared code.

Reference definitions used (file:line under /root/reference):
  residual / Jacobian blocks   src/factors/projection_factor.rs:184-296
  BAL camera                   crates/apex-camera-models/src/bal_pinhole.rs:273-296,400-435,528-556,649-672
  Huber corrector              src/core/corrector.rs:143-181 ; src/core/loss_functions.rs:364-380
  damped normal equations      src/linalg/sparse/explicit_schur.rs:1129-1234 (lambda*I damping)
  retraction                   crates/apex-manifolds/src/se3.rs:569-583, so3.rs:558-612
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def quat_to_R(q):
    """(n,4) (w,x,y,z) (not necessarily unit) -> (n,3,3) rotation of the normalised quaternion."""
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.empty((q.shape[0], 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - w * z); R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y); R[:, 2, 1] = 2 * (y * z + w * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def skew(v):
    n = v.shape[0]
    S = np.zeros((n, 3, 3))
    S[:, 0, 1] = -v[:, 2]; S[:, 0, 2] = v[:, 1]
    S[:, 1, 0] = v[:, 2]; S[:, 1, 2] = -v[:, 0]
    S[:, 2, 0] = -v[:, 1]; S[:, 2, 1] = v[:, 0]
    return S


def so3_exp_R(theta):
    """Rodrigues, (n,3) -> (n,3,3)."""
    ang = np.linalg.norm(theta, axis=-1)
    K = skew(theta)
    small = ang < 1e-8
    a = np.where(small, 1.0 - ang**2 / 6, np.sin(ang) / np.where(small, 1, ang))
    b = np.where(small, 0.5 - ang**2 / 24, (1 - np.cos(ang)) / np.where(small, 1, ang**2))
    return np.eye(3)[None] + a[:, None, None] * K + b[:, None, None] * (K @ K)


def so3_V(theta):
    ang = np.linalg.norm(theta, axis=-1)
    K = skew(theta)
    small = ang < 1e-6
    b = np.where(small, 0.5 - ang**2 / 24, (1 - np.cos(ang)) / np.where(small, 1, ang**2))
    c = np.where(small, 1.0 / 6 - ang**2 / 120, (ang - np.sin(ang)) / np.where(small, 1, ang**3))
    return np.eye(3)[None] + b[:, None, None] * K + c[:, None, None] * (K @ K)


def project(poses, intr, pts, cam_idx, pt_idx):
    """Raw (uncorrected) projection and validity per observation."""
    R = quat_to_R(poses[:, 3:7])[cam_idx]
    pc = np.einsum("nij,nj->ni", R, pts[pt_idx]) + poses[cam_idx, 0:3]
    valid = pc[:, 2] < -1e-6
    z = np.where(valid, pc[:, 2], -1.0)
    xn, yn = -pc[:, 0] / z, -pc[:, 1] / z
    r2 = xn * xn + yn * yn
    f, k1, k2 = intr[cam_idx, 0], intr[cam_idx, 1], intr[cam_idx, 2]
    d = 1 + k1 * r2 + k2 * r2 * r2
    return np.stack([f * xn * d, f * yn * d], -1), valid, pc, R


def residuals(poses, intr, pts, cam_idx, pt_idx, obs_uv, huber_delta=1.0):
    """Huber-corrected residual (n_obs,2) and cost = 1/2 |r~|^2."""
    uv, valid, _, _ = project(poses, intr, pts, cam_idx, pt_idx)
    r = (uv - obs_uv) * valid[:, None]
    w = np.ones(r.shape[0])
    if huber_delta > 0:
        s = np.sum(r * r, -1)
        out = s > huber_delta**2
        w = np.where(out, np.sqrt(huber_delta / np.sqrt(np.where(out, s, 1.0))), 1.0)
    rt = r * w[:, None]
    return rt, 0.5 * float(np.sum(rt * rt)), w, valid


def jacobian_blocks(poses, intr, pts, cam_idx, pt_idx, obs_uv, huber_delta=1.0):
    """Analytic blocks (corrected): Jpose (n,2,6), Jpt (n,2,3), Jintr (n,2,3)."""
    rt, cost, w, valid = residuals(poses, intr, pts, cam_idx, pt_idx, obs_uv, huber_delta)
    _, _, pc, R = project(poses, intr, pts, cam_idx, pt_idx)
    f, k1, k2 = intr[cam_idx, 0], intr[cam_idx, 1], intr[cam_idx, 2]
    z = np.where(valid, pc[:, 2], -1.0)
    X, Y = pc[:, 0], pc[:, 1]
    xn, yn = -X / z, -Y / z
    r2 = xn**2 + yn**2
    d = 1 + k1 * r2 + k2 * r2**2
    dp = k1 + 2 * k2 * r2  # d'(r2)
    # d(xn,yn)/d(X,Y,Z)
    dn = np.zeros((len(z), 2, 3))
    dn[:, 0, 0] = -1 / z; dn[:, 0, 2] = X / z**2
    dn[:, 1, 1] = -1 / z; dn[:, 1, 2] = Y / z**2
    # d(u,v)/d(xn,yn) = f (d I + 2 d' n n^T)
    n = np.stack([xn, yn], -1)
    du = f[:, None, None] * (d[:, None, None] * np.eye(2)[None] + 2 * dp[:, None, None] * n[:, :, None] * n[:, None, :])
    Jpc = du @ dn
    pw = pts[pt_idx]
    dpc = np.concatenate([R, -R @ skew(pw)], axis=2)  # right perturbation [rho, theta]
    Jpose = Jpc @ dpc
    Jpt = Jpc @ R
    Jintr = np.stack([
        np.stack([xn * d, f * xn * r2, f * xn * r2**2], -1),
        np.stack([yn * d, f * yn * r2, f * yn * r2**2], -1)], 1)
    sc = (w * valid)[:, None, None]
    return rt, cost, Jpose * sc, Jpt * sc, Jintr * sc


def sparse_jacobian(Jpose, Jpt, Jintr, cam_idx, pt_idx, lay, selfcal=True):
    """J in the reference's global column order (rows: observation insertion order)."""
    n = Jpose.shape[0]
    rows = np.arange(2 * n).reshape(n, 2)
    R, Cc, V = [], [], []

    def add(blk, col0):
        w = blk.shape[2]
        R.append(np.repeat(rows[:, :, None], w, 2).ravel())
        Cc.append(np.repeat((col0[:, None] + np.arange(w)[None])[:, None, :], 2, 1).ravel())
        V.append(blk.ravel())

    add(Jpose, lay.pose_col[cam_idx])
    add(Jpt, lay.pt_col[pt_idx])
    if selfcal:
        add(Jintr, lay.intr_col[cam_idx])
    return sp.csc_matrix((np.concatenate(V), (np.concatenate(R), np.concatenate(Cc))), shape=(2 * n, lay.total_dof))


def direct_step(J, r, lam):
    """Solve (J^T J + lam I) dx = -J^T r directly (no Schur)."""
    H = (J.T @ J).tocsc()
    g = J.T @ r
    A = H + lam * sp.identity(H.shape[0], format="csc")
    n = A.shape[0]
    if n <= 6000:
        dx = np.linalg.solve(A.toarray(), -g)
    else:
        dx = spla.spsolve(A, -g)
    return dx, g, H


def schur_dense(H, g, cam_dof, lam):
    """Dense S and reduced rhs from the full H (reference column order: cameras first)."""
    Hd = H.toarray() if sp.issparse(H) else H
    Hcc = Hd[:cam_dof, :cam_dof] + lam * np.eye(cam_dof)
    Hcl = Hd[:cam_dof, cam_dof:]
    Hll = Hd[cam_dof:, cam_dof:] + lam * np.eye(Hd.shape[0] - cam_dof)
    # per-landmark 3x3 inversion with the reference's eigenvalue gate
    # (explicit_schur.rs:377-442, called with lambda = 0.0): min_ev < 1e-12 ->
    # + (1e-6 + max_ev*1e-6) I ; cond > 1e10 -> + max_ev*1e-6 I ; else plain inverse.
    n_l = Hll.shape[0] // 3
    Hll_inv = np.zeros_like(Hll)
    for l in range(n_l):
        B = Hll[3 * l:3 * l + 3, 3 * l:3 * l + 3]
        ev = np.linalg.eigvalsh(B)
        if ev[0] < 1e-12:
            B = B + (1e-6 + ev[-1] * 1e-6) * np.eye(3)
        elif ev[-1] / ev[0] > 1e10:
            B = B + ev[-1] * 1e-6 * np.eye(3)
        Hll_inv[3 * l:3 * l + 3, 3 * l:3 * l + 3] = np.linalg.inv(B)
    E = Hcl @ Hll_inv @ Hcl.T
    S = Hcc - E
    e = Hcl @ (Hll_inv @ (-g[cam_dof:]))
    gred = -g[:cam_dof] - e
    schur_dense.last_gscale = float(max(np.abs(g[:cam_dof]).max(), np.abs(e).max()))
    schur_dense.last_scale = float(max(np.abs(Hcc).max(), np.abs(E).max()))  # size of the cancelled terms
    return S, gred


def retract(poses, intr, pts, step, lay, fix_pose=None):
    """x (+) step with the right-plus SE3 retraction; returns new (poses, intr, pts).
    Quaternions are returned normalised (the reference stores the raw product and
    normalises on the next use; the rotation is the same)."""
    n_cam = poses.shape[0]
    d = step[lay.pose_col[:, None] + np.arange(6)[None]].copy()
    if fix_pose is not None:
        d[fix_pose.astype(bool)] = 0.0
    R = quat_to_R(poses[:, 3:7])
    Rn = R @ so3_exp_R(d[:, 3:6])
    tn = poses[:, 0:3] + np.einsum("nij,nj->ni", R, np.einsum("nij,nj->ni", so3_V(d[:, 3:6]), d[:, 0:3]))
    from scipy.spatial.transform import Rotation

    qxyzw = Rotation.from_matrix(Rn).as_quat()
    q = np.concatenate([qxyzw[:, 3:4], qxyzw[:, 0:3]], -1)
    # keep the hemisphere of the input quaternion
    sgn = np.sign(np.sum(q * poses[:, 3:7], -1))
    sgn[sgn == 0] = 1
    q *= sgn[:, None]
    new_poses = np.concatenate([tn, q], -1)
    new_intr = intr + step[lay.intr_col[:, None] + np.arange(3)[None]]
    new_pts = pts + step[lay.pt_col[:, None] + np.arange(3)[None]]
    return new_poses, new_intr, new_pts
