"""BAL datasets: Python mirror of `apex_io::BalLoader` (crates/apex-io/src/bal.rs:33-202) over the
library's C++ reader (csrc/bal_io.cpp), plus the problem construction of
bin/bundle_adjustment.rs:200-257.  Host code only -- works without a GPU."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import capi
from .synthetic import BAProblemData

DEFAULT_FOCAL_LENGTH = 500.0  # bal.rs:98

IO_ERROR_NAMES = {-20: "Io", -21: "Parse", -22: "MissingFields", -23: "InvalidNumber", -5: "InvalidInput"}


class IoError(RuntimeError):
    """Mirror of apex_io::IoError: `.kind` in {Io, Parse, MissingFields, InvalidNumber}."""

    def __init__(self, code: int, message: str):
        self.code = code
        self.kind = IO_ERROR_NAMES.get(code, f"Error({code})")
        super().__init__(f"{self.kind}: {message}")


@dataclass
class BalDataset:
    """bal.rs:85-93 with the arrays kept columnar.
    cameras[:, 0:3] rotation (axis-angle), [:, 3:6] translation, [:, 6] focal_length, [:, 7:9] k1 k2."""

    cameras: np.ndarray        # (n_cam, 9)
    points: np.ndarray         # (n_pt, 3)
    camera_index: np.ndarray   # (n_obs,) uint32
    point_index: np.ndarray    # (n_obs,) uint32
    observations: np.ndarray   # (n_obs, 2) pixel x, y
    poses: np.ndarray          # (n_cam, 7) [t, qw,qx,qy,qz] as run_bundle_adjustment builds them
    intrinsics: np.ndarray     # (n_cam, 3) [f, k1, k2]

    def to_problem_data(self, num_points: int | None = None, name: str = "bal") -> BAProblemData:
        """bin/bundle_adjustment.rs:172-173, 260-265: optionally keep the first `num_points` landmarks
        and the observations that reference them."""
        n_pt = self.points.shape[0] if num_points is None else min(int(num_points), self.points.shape[0])
        keep = self.point_index < n_pt
        return BAProblemData(poses=self.poses.copy(), intr=self.intrinsics.copy(), points=self.points[:n_pt].copy(),
                             cam_idx=np.ascontiguousarray(self.camera_index[keep]),
                             pt_idx=np.ascontiguousarray(self.point_index[keep]),
                             obs_uv=np.ascontiguousarray(self.observations[keep]), name=name)


class BalLoader:
    @staticmethod
    def load(path) -> BalDataset:
        L = capi.load()
        h = C.c_void_p()
        rc = L.apexgpu_bal_open(str(path).encode(), C.byref(h))
        if rc != 0:
            raise IoError(rc, L.apexgpu_bal_last_error().decode())
        try:
            nc, npt, no = C.c_int64(), C.c_int64(), C.c_int64()
            L.apexgpu_bal_sizes(h, C.byref(nc), C.byref(npt), C.byref(no))
            cam_idx = np.empty(no.value, dtype=np.uint32); pt_idx = np.empty(no.value, dtype=np.uint32)
            uv = np.empty((no.value, 2)); cams = np.empty((nc.value, 9)); pts = np.empty((npt.value, 3))
            L.apexgpu_bal_raw(h, capi.ptr(cam_idx), capi.ptr(pt_idx), capi.ptr(uv), capi.ptr(cams), capi.ptr(pts))
            poses = np.empty((nc.value, 7)); intr = np.empty((nc.value, 3))
            L.apexgpu_bal_variables(h, capi.ptr(poses), capi.ptr(intr))
        finally:
            L.apexgpu_bal_close(h)
        if no.value and (cam_idx.max(initial=0) >= nc.value or pt_idx.max(initial=0) >= npt.value):
            # the reference would panic on the out-of-range index when it builds the factors
            raise IoError(-21, "observation references a camera or point beyond the header counts")
        return BalDataset(cams, pts, cam_idx, pt_idx, uv, poses, intr)


def write_bal(path, data: BAProblemData):
    """Write a problem in BAL text form (rotation as axis-angle); the inverse of the loader, used by
    tests to round-trip synthetic problems."""
    q = data.poses[:, 3:7] / np.linalg.norm(data.poses[:, 3:7], axis=1, keepdims=True)
    w = np.clip(q[:, 0], -1.0, 1.0)
    sgn = np.where(w < 0, -1.0, 1.0)
    q = q * sgn[:, None]
    ang = 2.0 * np.arctan2(np.linalg.norm(q[:, 1:], axis=1), q[:, 0])
    sh = np.linalg.norm(q[:, 1:], axis=1)
    aa = np.where(sh[:, None] > 1e-300, q[:, 1:] / np.maximum(sh, 1e-300)[:, None] * ang[:, None], 0.0)
    with open(path, "w") as f:
        f.write(f"{data.n_cam} {data.n_pt} {data.n_obs}\n")
        for c, p, (x, y) in zip(data.cam_idx, data.pt_idx, data.obs_uv):
            f.write(f"{int(c)} {int(p)}     {float(x)!r} {float(y)!r}\n")
        for i in range(data.n_cam):
            for v in (*aa[i], *data.poses[i, :3], *data.intr[i]):
                f.write(f"{float(v)!r}\n")
        for j in range(data.n_pt):
            for v in data.points[j]:
                f.write(f"{float(v)!r}\n")


def reference_columns(n_cam: int, n_pt: int):
    """The library's (C++) version of layout.reference_column_layout."""
    L = capi.load()
    ic = np.empty(n_cam, dtype=np.int64); pc = np.empty(n_cam, dtype=np.int64); tc = np.empty(n_pt, dtype=np.int64)
    rc = L.apexgpu_reference_columns(n_cam, n_pt, capi.ptr(ic), capi.ptr(pc), capi.ptr(tc))
    if rc != 0:
        raise capi.LinAlgError(rc, "apexgpu_reference_columns")
    return ic, pc, tc
