"""Reference global column layout.

The reference orders Jacobian columns by the LEXICOGRAPHIC order of the variable
names (src/optimizer/mod.rs:530-536).  The bundle-adjustment callers name their
variables `pose_{:04}`, `intr_{:04}`, `pt_{:05}` (bin/bundle_adjustment.rs:232-257),
so the global layout is [all intr_* | all pose_* | all pt_*] with, inside each
family, the byte order of the zero-padded decimal strings -- which stops being the
numeric order once an index outgrows its pad width (`pt_100000` sorts between
`pt_10000` and `pt_10001`).  The device keeps a numeric, camera-major layout; these
offsets are what `apexgpu_set_structure` uses to permute at the boundary.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


def _lex_rank(n: int, width: int) -> np.ndarray:
    """rank[i] = position of the name with index i among n names `{i:0{width}d}`
    sorted as byte strings."""
    if n <= 10**width:
        return np.arange(n, dtype=np.int64)
    names = [f"{i:0{width}d}" for i in range(n)]
    order = sorted(range(n), key=names.__getitem__)
    rank = np.empty(n, dtype=np.int64)
    rank[np.asarray(order, dtype=np.int64)] = np.arange(n, dtype=np.int64)
    return rank


@dataclass
class ColumnLayout:
    intr_col: np.ndarray  # (n_cam,) first global column of intr_i   (3 columns)
    pose_col: np.ndarray  # (n_cam,) first global column of pose_i   (6 columns)
    pt_col: np.ndarray  # (n_pt,)  first global column of pt_j     (3 columns)
    cam_dof: int  # 9*n_cam : camera-side columns come first
    total_dof: int


def reference_column_layout(n_cam: int, n_pt: int) -> ColumnLayout:
    cam_rank = _lex_rank(n_cam, 4)
    pt_rank = _lex_rank(n_pt, 5)
    intr_col = 3 * cam_rank
    pose_col = 3 * n_cam + 6 * cam_rank
    pt_col = 9 * n_cam + 3 * pt_rank
    return ColumnLayout(
        intr_col=intr_col.astype(np.int64),
        pose_col=pose_col.astype(np.int64),
        pt_col=pt_col.astype(np.int64),
        cam_dof=9 * n_cam,
        total_dof=9 * n_cam + 3 * n_pt,
    )
