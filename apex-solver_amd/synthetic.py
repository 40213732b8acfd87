"""Deterministic synthetic bundle-adjustment problems.

Counter-based SplitMix64 streams (seed base 0xA9E50000 + config_id) so that every
array element is a pure function of (seed, stream, index): the generator is
vectorised, reproducible across machines, and any sub-range can be regenerated
without the rest.

Geometry: cameras on a ring of radius 10 looking at the origin in the BAL -Z
convention (p_cam = R p_w + t, the scene sits near p_cam.z = -10 < -1e-6, the
validity limit of the reference's BALPinholeCameraStrict,
crates/apex-camera-models/src/bal_pinhole.rs:154-156); landmarks uniform in
[-2,2]^3; landmark j is seen by k_j distinct cameras out of a window of 64
consecutive ring cameras centred at floor(j*N_cam/N_pt), which gives the banded
reduced camera matrix of a sequential capture.

The parameter conventions are the reference's (bin/bundle_adjustment.rs:232-257):
pose = [tx,ty,tz,qw,qx,qy,qz], intrinsics = [f,k1,k2], point = [x,y,z].
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_STREAM = np.uint64(0x632BE59BD9B4E019)
SEED_BASE = 0xA9E50000

#: The five BASELINE.json shapes: name -> (config_id, n_cam, n_pt, k_lo, k_hi)
#: k_j ~ U{k_lo..k_hi}; (k_lo+k_hi)/2 matches N_obs/N_pt of the real dataset.
SHAPES = {
    "ladybug-49": (0, 49, 7776, 3, 5),
    "ladybug-1723": (2, 1723, 156502, 3, 6),
    "venice-1778": (3, 1778, 993923, 3, 7),
    "final-13682": (4, 13682, 4456117, 3, 10),
    "synthetic-10k": (5, 10000, 2000000, 3, 9),
}


def _mix64(z: np.ndarray) -> np.ndarray:
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


class SplitMix:
    """Counter-based SplitMix64: u64(stream, i) = mix(seed + stream*C1 + (i+1)*golden)."""

    def __init__(self, seed: int):
        self.seed = np.uint64(seed & 0xFFFFFFFFFFFFFFFF)

    def u64(self, stream: int, n: int, offset: int = 0) -> np.ndarray:
        with np.errstate(over="ignore"):
            i = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
            z = self.seed + np.uint64(stream) * _STREAM + i * _GOLDEN
            return _mix64(z)

    def uniform(self, stream: int, n: int, offset: int = 0) -> np.ndarray:
        """U[0,1) with 53 random bits."""
        return (self.u64(stream, n, offset) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)

    def normal(self, stream: int, n: int, offset: int = 0) -> np.ndarray:
        """N(0,1) by Box-Muller on streams (stream, stream+1)."""
        u1 = self.uniform(stream, n, offset)
        u2 = self.uniform(stream + 1, n, offset)
        return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)


def quat_mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Hamilton product, (w,x,y,z) rows."""
    aw, ax, ay, az = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bw, bx, by, bz = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack(
        [
            aw * bw - ax * bx - ay * by - az * bz,
            aw * bx + ax * bw + ay * bz - az * by,
            aw * by - ax * bz + ay * bw + az * bx,
            aw * bz + ax * by - ay * bx + az * bw,
        ],
        axis=-1,
    )


def quat_exp(theta: np.ndarray) -> np.ndarray:
    """Unit quaternion of the rotation vector theta (n,3)."""
    ang = np.linalg.norm(theta, axis=-1)
    half = 0.5 * ang
    s = np.where(ang > 1e-12, np.sin(half) / np.where(ang > 1e-12, ang, 1.0), 0.5)
    return np.concatenate([np.cos(half)[..., None], theta * s[..., None]], axis=-1)


def quat_to_rot(q: np.ndarray) -> np.ndarray:
    """(n,4) unit quaternions (w,x,y,z) -> (n,9) row-major rotation matrices."""
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return np.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], axis=-1)


def quat_rotate(q: np.ndarray, v: np.ndarray) -> np.ndarray:
    """q v q* for unit quaternions (rows broadcast)."""
    qv = q[..., 1:]
    t = 2.0 * np.cross(qv, v)
    return v + q[..., :1] * t + np.cross(qv, t)


def project_bal(poses: np.ndarray, intr: np.ndarray, points: np.ndarray) -> np.ndarray:
    """BAL projection of matched rows (poses n×7, intr n×3, points n×3) → (n,2).
    Formula of bal_pinhole.rs:273-296 (used only to synthesise observations)."""
    pc = quat_rotate(poses[:, 3:7], points) + poses[:, 0:3]
    inz = -1.0 / pc[:, 2]
    xn = pc[:, 0] * inz
    yn = pc[:, 1] * inz
    r2 = xn * xn + yn * yn
    d = 1.0 + intr[:, 1] * r2 + intr[:, 2] * r2 * r2
    return np.stack([intr[:, 0] * xn * d, intr[:, 0] * yn * d], axis=-1)


@dataclass
class BAProblemData:
    """A bundle-adjustment problem in the reference's parameter conventions.

    Observations are in file order (not sorted); cam_idx/pt_idx index the camera
    and landmark arrays.  `truth_*` are only set by the synthetic generator.
    """

    poses: np.ndarray  # (n_cam,7) tx,ty,tz,qw,qx,qy,qz
    intr: np.ndarray  # (n_cam,3) f,k1,k2
    points: np.ndarray  # (n_pt,3)
    cam_idx: np.ndarray  # (n_obs,) uint32
    pt_idx: np.ndarray  # (n_obs,) uint32
    obs_uv: np.ndarray  # (n_obs,2)
    name: str = "custom"
    truth_poses: np.ndarray | None = None
    truth_intr: np.ndarray | None = None
    truth_points: np.ndarray | None = None

    @property
    def n_cam(self) -> int:
        return int(self.poses.shape[0])

    @property
    def n_pt(self) -> int:
        return int(self.points.shape[0])

    @property
    def n_obs(self) -> int:
        return int(self.cam_idx.shape[0])


def make_problem(
    n_cam: int,
    n_pt: int,
    k_lo: int = 3,
    k_hi: int = 9,
    config_id: int = 99,
    window: int = 64,
    outlier_frac: float = 0.02,
    name: str = "synthetic",
    behind_frac: float = 0.0,
    hub_frac: float = 0.0,
    hub_obs_prob: float = 0.25,
    long_range_prob: float = 0.0001,
    mix_frac: float = 0.0,
) -> BAProblemData:
    """Generate a synthetic BA problem (see module docstring).

    `behind_frac` > 0 moves that fraction of the landmarks behind a camera's
    image plane so that the cheirality branch (zero residual and Jacobian,
    projection_factor.rs:227-238) is exercised.

    `mix_frac` > 0 is the structure sweep between "banded" and "hubs" ("<shape>-mix:<p>"): that fraction of the landmarks
    ignores the capture window and draws each of its cameras from a global POPULARITY law over all cameras -- rank r with
    probability ~ 1 / (r + 1), ranks dealt to cameras by a fixed pseudo-random permutation -- the co-visibility of an
    internet photo collection (crates/apex-io/datasets.toml:155-156: final-13682 is one) rather than of a capture
    sequence.  A landmark may draw a camera twice (duplicate observations are legal input: tests/test_gpu_parity.py).

    `hub_frac` > 0 is the NON-BANDED stress variant ("<shape>-hub"): that fraction of the cameras (evenly spread over
    the ring) are hubs; a landmark anywhere on the ring swaps one of its window cameras for a random hub with
    probability `hub_obs_prob` (so a hub is covisible with every camera, like the landmark-rich overview photographs of
    a community collection) and another one for a uniformly random camera with probability `long_range_prob` (random
    long-range pairs).  Sizes, observation counts and noise are those of the banded shape.
    """
    rng = SplitMix(SEED_BASE + config_id)
    W = min(window, n_cam)
    k_hi = min(k_hi, W)
    k_lo = min(k_lo, k_hi)

    # --- cameras -----------------------------------------------------------
    phi = 2.0 * np.pi * np.arange(n_cam) / n_cam
    q_z = np.stack([np.cos(-phi / 2), np.zeros(n_cam), np.zeros(n_cam), np.sin(-phi / 2)], axis=-1)
    q_p = np.broadcast_to(np.array([0.5, -0.5, -0.5, -0.5]), (n_cam, 4))
    jitter = 0.01 * np.stack([rng.normal(10 + 2 * a, n_cam) for a in range(3)], axis=-1)
    q_true = quat_mul(quat_mul(q_p, q_z), quat_exp(jitter))
    t_true = np.tile(np.array([0.0, 0.0, -10.0]), (n_cam, 1))
    truth_poses = np.concatenate([t_true, q_true], axis=-1)
    f = 400.0 + 800.0 * rng.uniform(20, n_cam)
    k1 = 1e-2 * rng.normal(21, n_cam)
    k2 = 1e-3 * rng.normal(23, n_cam)
    truth_intr = np.stack([f, k1, k2], axis=-1)

    # --- landmarks ---------------------------------------------------------
    truth_points = -2.0 + 4.0 * np.stack([rng.uniform(30 + a, n_pt) for a in range(3)], axis=-1)

    # --- visibility: k_j distinct cameras, one per stratum of the window ----
    kspan = k_hi - k_lo + 1
    k = (k_lo + np.floor(rng.uniform(40, n_pt) * kspan)).astype(np.int64)
    k = np.minimum(k, k_hi)
    centre = (np.arange(n_pt, dtype=np.int64) * n_cam) // n_pt
    pt_ptr = np.zeros(n_pt + 1, dtype=np.int64)
    np.cumsum(k, out=pt_ptr[1:])
    n_obs = int(pt_ptr[-1])
    pt_idx = np.repeat(np.arange(n_pt, dtype=np.int64), k)
    slot = np.arange(n_obs, dtype=np.int64) - pt_ptr[pt_idx]  # 0..k_j-1
    kk = k[pt_idx]
    stratum = W // kk
    off = slot * stratum + np.floor(rng.uniform(41, n_obs) * stratum).astype(np.int64)
    cam_idx = (centre[pt_idx] - W // 2 + off) % n_cam
    if mix_frac > 0.0:
        mixed = rng.uniform(46, n_pt) < mix_frac
        # popularity ranks by the inverse of P(rank <= r) = ln(r + 1) / ln(n + 1); the slot index is added so that one
        # landmark rarely draws a rank twice
        u = rng.uniform(47, n_obs)
        rank = np.minimum(np.floor(np.exp(u * np.log(n_cam + 1.0))).astype(np.int64) - 1 + slot, n_cam - 1)
        order = np.argsort(rng.uniform(48, n_cam), kind="stable")      # rank -> camera
        sel = mixed[pt_idx]
        cam_idx = np.where(sel, order[rank], cam_idx)
    if hub_frac > 0.0:
        n_hub = max(1, int(round(hub_frac * n_cam)))
        hubs = (np.arange(n_hub, dtype=np.int64) * n_cam) // n_hub + (n_cam // (2 * n_hub))
        first = pt_ptr[:-1]
        # slot 0 of a landmark -> a random hub (kept only when the hub is not one of the landmark's cameras already:
        # window cameras of slots >= 1 lie at least one stratum after slot 0's, so only slot 0's own camera can clash)
        take = rng.uniform(42, n_pt) < hub_obs_prob
        hub_pick = hubs[np.minimum((rng.uniform(43, n_pt) * n_hub).astype(np.int64), n_hub - 1)]
        lo = (centre - W // 2) % n_cam
        inside = ((hub_pick - lo) % n_cam) < W          # the hub sits inside this landmark's own window: leave it
        sel = take & ~inside
        cam_idx[first[sel]] = hub_pick[sel]
        # slot 1 -> any camera of the ring (a long-range pair), outside the window and not the hub just chosen
        take2 = (rng.uniform(44, n_pt) < long_range_prob) & (k >= 2)
        far = np.minimum((rng.uniform(45, n_pt) * n_cam).astype(np.int64), n_cam - 1)
        ok2 = take2 & (((far - lo) % n_cam) >= W) & (far != cam_idx[first])
        cam_idx[first[ok2] + 1] = far[ok2]

    # --- observations --------------------------------------------------------
    # exact BAL projection of the truth (bal_pinhole.rs:273-296), structure-of-arrays and through
    # per-camera rotation matrices: this loop is the generator's hot spot at 29 M observations
    R = quat_to_rot(q_true)  # (n_cam, 9)
    ci = cam_idx
    px, py, pz = (np.take(truth_points[:, a], pt_idx) for a in range(3))
    pc = []
    for r in range(3):
        acc = np.take(R[:, 3 * r], ci) * px
        acc += np.take(R[:, 3 * r + 1], ci) * py
        acc += np.take(R[:, 3 * r + 2], ci) * pz
        acc += np.take(t_true[:, r], ci)
        pc.append(acc)
    del px, py, pz
    inz = -1.0 / pc[2]
    xn = pc[0] * inz
    yn = pc[1] * inz
    del pc, inz
    r2 = xn * xn + yn * yn
    dist = 1.0 + np.take(k1, ci) * r2 + np.take(k2, ci) * (r2 * r2)
    del r2
    fd = np.take(f, ci) * dist
    del dist
    uv = np.empty((n_obs, 2))
    uv[:, 0] = fd * xn
    uv[:, 1] = fd * yn
    del fd, xn, yn
    uv[:, 0] += 0.5 * rng.normal(50, n_obs)
    uv[:, 1] += 0.5 * rng.normal(52, n_obs)
    is_out = rng.uniform(54, n_obs) < outlier_frac
    uv[:, 0] += is_out * (-20.0 + 40.0 * rng.uniform(55, n_obs))
    uv[:, 1] += is_out * (-20.0 + 40.0 * rng.uniform(56, n_obs))
    del is_out

    # --- initial parameters = truth + noise ---------------------------------
    points = truth_points + 1e-2 * np.stack([rng.normal(60 + 2 * a, n_pt) for a in range(3)], axis=-1)
    if behind_frac > 0.0:
        # push some landmarks far along +z of their first observing camera (behind it)
        sel = rng.uniform(70, n_pt) < behind_frac
        first_cam = cam_idx[pt_ptr[:-1]]
        # camera centre direction in world: R^T e3 ; moving 25 along it puts p_cam.z > 0
        qc = truth_poses[first_cam, 3:7] * np.array([1.0, -1.0, -1.0, -1.0])
        dirw = quat_rotate(qc, np.tile(np.array([0.0, 0.0, 1.0]), (n_pt, 1)))
        points = points + sel[:, None] * 25.0 * dirw
    dtan = 1e-3 * np.stack([rng.normal(80 + 2 * a, n_cam) for a in range(6)], axis=-1)
    # right-perturbation T*Exp(d) to first order is enough for an initial guess
    q0 = quat_mul(q_true, quat_exp(dtan[:, 3:6]))
    q0 /= np.linalg.norm(q0, axis=-1, keepdims=True)
    t0 = t_true + quat_rotate(q_true, dtan[:, 0:3])
    poses = np.concatenate([t0, q0], axis=-1)
    intr = truth_intr.copy()
    intr[:, 0] *= 1.0 + 1e-3 * rng.normal(90, n_cam)

    return BAProblemData(
        poses=np.ascontiguousarray(poses),
        intr=np.ascontiguousarray(intr),
        points=np.ascontiguousarray(points),
        cam_idx=cam_idx.astype(np.uint32),
        pt_idx=pt_idx.astype(np.uint32),
        obs_uv=np.ascontiguousarray(uv),
        name=name,
        truth_poses=truth_poses,
        truth_intr=truth_intr,
        truth_points=truth_points,
    )


def make_named(shape: str, scale: float = 1.0) -> BAProblemData:
    """One of the BASELINE.json shapes; `scale` < 1 shrinks cameras and landmarks
    proportionally (same generator, same per-landmark statistics)."""
    hub = shape.endswith("-hub")
    mix = 0.0
    base = shape[:-4] if hub else shape
    if "-mix:" in base:
        base, frac = base.split("-mix:")
        mix = float(frac)
    cid, n_cam, n_pt, k_lo, k_hi = SHAPES[base]
    if scale != 1.0:
        n_cam = max(8, int(round(n_cam * scale)))
        n_pt = max(16, int(round(n_pt * scale)))
    nm = shape if scale == 1.0 else f"{shape}@{scale:g}"
    # Generating 29 M observations takes about a minute of numpy: tools that run several processes on the same box
    # (profiles, A/B benches) may share one copy through APEX_SYNTH_CACHE=<directory>.  Pure cache: same arrays.
    import os
    cache = os.environ.get("APEX_SYNTH_CACHE")
    path = os.path.join(cache, f"{nm}.npz") if cache else None
    if path and os.path.exists(path):
        z = np.load(path)
        return BAProblemData(**{k: z[k] for k in z.files}, name=nm)
    d = _make_named_uncached(shape, scale, nm, hub, cid, n_cam, n_pt, k_lo, k_hi, mix)
    if path:
        os.makedirs(cache, exist_ok=True)
        tmp = path + f".{os.getpid()}.tmp.npz"
        np.savez(tmp, poses=d.poses, intr=d.intr, points=d.points, cam_idx=d.cam_idx, pt_idx=d.pt_idx, obs_uv=d.obs_uv,
                 truth_poses=d.truth_poses, truth_intr=d.truth_intr, truth_points=d.truth_points)
        os.replace(tmp, path)
    return d


def _make_named_uncached(shape, scale, nm, hub, cid, n_cam, n_pt, k_lo, k_hi, mix=0.0) -> BAProblemData:
    return make_problem(n_cam, n_pt, k_lo, k_hi, config_id=cid, name=nm, hub_frac=0.015 if hub else 0.0, mix_frac=mix)


# ---------------------------------------------------------------------------------------------------
# SE3 pose graphs (BASELINE.json configs[1]; SURVEY.md §8(d): "50x50 sphere of SE3 poses, odometry +
# loop edges (2,500 v / 4,949 e), noise N(0, 0.05^2) on the tangent")
# ---------------------------------------------------------------------------------------------------
@dataclass
class PoseGraphData:
    """What bin/pose_graph_g2o.rs:748-830 feeds the optimiser: vertices sorted by id with
    pose = [tx,ty,tz,qw,qx,qy,qz]; edge e is BetweenFactor(meas[e]) on (x{ids[e_from[e]]}, x{ids[e_to[e]]})."""

    ids: np.ndarray       # (n_v,) int64, ascending
    poses: np.ndarray     # (n_v, 7) initial values
    e_from: np.ndarray    # (n_e,) uint32 index into ids
    e_to: np.ndarray      # (n_e,) uint32
    meas: np.ndarray      # (n_e, 7)
    truth: np.ndarray | None = None
    name: str = "pose-graph"

    @property
    def n_v(self) -> int:
        return int(self.poses.shape[0])

    @property
    def n_e(self) -> int:
        return int(self.meas.shape[0])


def se3_exp(tau: np.ndarray) -> np.ndarray:
    """Exp of (n,6) tangents [rho, theta] -> (n,7) poses (generator-side helper)."""
    th = tau[:, 3:6]
    ang = np.linalg.norm(th, axis=-1)
    small = ang < 1e-8
    a = np.where(small, 1.0, ang)
    c1 = np.where(small, 0.5, (1.0 - np.cos(a)) / (a * a))
    c2 = np.where(small, 1.0 / 6.0, (a - np.sin(a)) / (a * a * a))
    k1 = np.cross(th, tau[:, 0:3])
    k2 = np.cross(th, k1)
    t = tau[:, 0:3] + c1[:, None] * k1 + c2[:, None] * k2
    return np.concatenate([t, quat_exp(th)], axis=-1)


def se3_mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    return np.concatenate([quat_rotate(a[:, 3:7], b[:, 0:3]) + a[:, 0:3], quat_mul(a[:, 3:7], b[:, 3:7])], axis=-1)


def se3_inv(a: np.ndarray) -> np.ndarray:
    qi = a[:, 3:7] * np.array([1.0, -1.0, -1.0, -1.0])
    return np.concatenate([-quat_rotate(qi, a[:, 0:3]), qi], axis=-1)


def make_sphere(rings: int = 50, per_ring: int = 50, noise: float = 0.05, radius: float = 50.0, config_id: int = 1,
                id_stride: int = 1) -> PoseGraphData:
    """sphere2500-shaped pose graph: rings*per_ring poses spiralling up a sphere, an odometry edge
    i -> i+1 for every consecutive pair and a loop-closure edge i -> i+per_ring between neighbouring
    rings (2,499 + 2,450 = 4,949 edges for 50x50).  Measurements are the true relative poses times
    Exp(N(0, noise^2)); initial values are the odometry chain from the true first pose (drifted)."""
    n = rings * per_ring
    rng = SplitMix(SEED_BASE + config_id)
    i = np.arange(n, dtype=np.float64)
    az = 2.0 * np.pi * (i % per_ring) / per_ring
    el = -0.45 * np.pi + 0.9 * np.pi * i / max(n - 1, 1)
    pos = radius * np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], axis=-1)
    # heading along the ring, small tilt: yaw about z then pitch about the local y axis
    yaw = az + 0.5 * np.pi
    qz = np.stack([np.cos(0.5 * yaw), 0 * yaw, 0 * yaw, np.sin(0.5 * yaw)], axis=-1)
    qy = np.stack([np.cos(0.5 * el), 0 * el, np.sin(0.5 * el), 0 * el], axis=-1)
    q = quat_mul(qz, qy)
    truth = np.concatenate([pos, q], axis=-1)
    ef = np.concatenate([np.arange(n - 1), np.arange(n - per_ring)]).astype(np.uint32)
    et = np.concatenate([np.arange(1, n), np.arange(per_ring, n)]).astype(np.uint32)
    m = ef.shape[0]
    rel = se3_mul(se3_inv(truth[ef]), truth[et])
    tau = noise * np.stack([rng.normal(10 + 2 * k, m) for k in range(6)], axis=-1)
    tau[:, 3:6] *= 0.2  # rotational noise in radians: a fifth of the translational sigma
    meas = se3_mul(rel, se3_exp(tau))
    meas[:, 3:7] /= np.linalg.norm(meas[:, 3:7], axis=-1, keepdims=True)
    init = np.empty_like(truth)
    init[0] = truth[0]
    for k in range(n - 1):  # odometry edges are the first n-1, edge k is k -> k+1
        init[k + 1] = se3_mul(init[k:k + 1], meas[k:k + 1])[0]
    init[:, 3:7] /= np.linalg.norm(init[:, 3:7], axis=-1, keepdims=True)
    ids = np.arange(n, dtype=np.int64) * id_stride
    return PoseGraphData(ids=ids, poses=init, e_from=ef, e_to=et, meas=meas, truth=truth, name=f"sphere-{n}")
