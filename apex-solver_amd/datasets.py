"""Real input files in the reference's on-disk layout, with the synthetic generator as the fallback.

The reference keeps its datasets where its downloader puts them (crates/apex-io/src/lib.rs:34-43, utils.rs:140-147,
186-189, 209-216):

    data/bundle_adjustment/{name}/problem-{cameras}-{points}-pre.txt      (BAL, bz2 archives unpacked by the downloader)
    data/odometry/3d/{file}.g2o                                           (e.g. sphere2500.g2o)

`load_named` / `load_pose_graph` look there first (relative to $APEX_DATA_ROOT, the working directory, then the repository
root), read what they find through the library's own readers (apexgpu_bal_open / apexgpu_g2o_open: csrc/bal_io.cpp,
g2o_io.cpp) and say so ("real"); otherwise they return the seeded synthetic problem of the same name ("synthetic").
A `.txt.bz2` next to the expected `.txt` (a download the reference has not unpacked yet) is accepted too.
Host code only.
"""
from __future__ import annotations

import bz2
import os
import tempfile

from . import synthetic

# BASELINE.json shape -> (registry name, cameras, points) of the BAL problem it stands for
BAL_FILES = {
    "ladybug-49": ("ladybug", 49, 7776),
    "ladybug-1723": ("ladybug", 1723, 156502),
    "venice-1778": ("venice", 1778, 993923),
    "final-13682": ("final", 13682, 4456117),
}
G2O_FILES = {"sphere2500": ("3d", "sphere2500.g2o")}

_REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def data_roots() -> list[str]:
    roots = []
    for r in (os.environ.get("APEX_DATA_ROOT"), os.getcwd(), _REPO_ROOT):
        if r and r not in roots:
            roots.append(r)
    return roots


def bal_path(shape: str) -> str | None:
    """The file the reference would read for this shape (ensure_ba_dataset, utils.rs:209-216), if it is on disk."""
    if shape not in BAL_FILES:
        return None
    name, cams, pts = BAL_FILES[shape]
    rel = os.path.join("data", "bundle_adjustment", name, f"problem-{cams}-{pts}-pre.txt")
    for root in data_roots():
        for cand in (os.path.join(root, rel), os.path.join(root, rel + ".bz2")):
            if os.path.isfile(cand):
                return cand
    return None


def g2o_path(shape: str) -> str | None:
    if shape not in G2O_FILES:
        return None
    cat, fn = G2O_FILES[shape]
    for root in data_roots():
        cand = os.path.join(root, "data", "odometry", cat, fn)
        if os.path.isfile(cand):
            return cand
    return None


def _open_maybe_bz2(path: str, loader):
    if not path.endswith(".bz2"):
        return loader(path)
    with tempfile.NamedTemporaryFile(suffix=".txt", delete=True) as tmp:   # decompress_bzip2 (utils.rs:238-242)
        with bz2.open(path, "rb") as src:
            for chunk in iter(lambda: src.read(1 << 24), b""):
                tmp.write(chunk)
        tmp.flush()
        return loader(tmp.name)


def load_named(shape: str, scale: float = 1.0):
    """-> (BAProblemData, "real" | "synthetic", source path or None).  A real file is used at scale 1 only (the scaled
    shapes are parity-test sizes of the generator)."""
    if scale == 1.0 and not shape.endswith("-hub") and "-mix:" not in shape:
        p = bal_path(shape)
        if p:
            from .bal import BalLoader

            ds = _open_maybe_bz2(p, BalLoader.load)
            return ds.to_problem_data(name=shape), "real", p
    return synthetic.make_named(shape, scale), "synthetic", None


def load_pose_graph(shape: str, rings: int = 50, per_ring: int = 50):
    """-> (PoseGraphData, "real" | "synthetic", source path or None)."""
    p = g2o_path(shape)
    if p:
        from .pose_graph import G2oLoader

        return G2oLoader.load(p).to_problem_data(name=shape), "real", p
    return synthetic.make_sphere(rings, per_ring), "synthetic", None
