"""Real input files are picked up from the reference's on-disk layout (crates/apex-io/src/utils.rs:140-147, 186-189,
209-216) when they are there; the seeded synthetic shapes stand in otherwise.  CPU only."""
import bz2
import os

import numpy as np
import pytest

import apex_solver_amd as pkg
from apex_solver_amd import datasets
from apex_solver_amd.bal import BalLoader, write_bal
from apex_solver_amd.pose_graph import write_g2o


@pytest.fixture()
def data_root(tmp_path, monkeypatch):
    monkeypatch.setenv("APEX_DATA_ROOT", str(tmp_path))
    monkeypatch.chdir(tmp_path)
    return tmp_path


def test_no_file_means_synthetic(data_root):
    assert datasets.bal_path("ladybug-49") is None
    d, kind, src = datasets.load_named("ladybug-49", 0.25)
    assert kind == "synthetic" and src is None and d.n_cam == 12


@pytest.mark.parametrize("packed", [False, True], ids=["txt", "txt.bz2"])
def test_bal_file_in_the_reference_layout_is_used(data_root, packed):
    small = pkg.synthetic.make_problem(7, 60, 3, 5, config_id=9)
    folder = data_root / "data" / "bundle_adjustment" / "ladybug"
    folder.mkdir(parents=True)
    txt = folder / "problem-49-7776-pre.txt"
    write_bal(txt, small)
    want = BalLoader.load(txt).to_problem_data()
    if packed:   # a download the reference has not unpacked yet (ensure_ba_dataset, utils.rs:231-246)
        (folder / "problem-49-7776-pre.txt.bz2").write_bytes(bz2.compress(txt.read_bytes()))
        txt.unlink()
    d, kind, src = datasets.load_named("ladybug-49")
    assert kind == "real" and src.startswith(str(folder)) and src.endswith(".bz2" if packed else ".txt")
    assert d.n_cam == 7 and d.n_pt == 60 and d.name == "ladybug-49"
    for a in ("poses", "intr", "points", "cam_idx", "pt_idx", "obs_uv"):
        assert np.array_equal(getattr(d, a), getattr(want, a)), a
    # a scaled shape is a parity-test size of the generator, never the file; other shapes are unaffected
    assert datasets.load_named("ladybug-49", 0.5)[1] == "synthetic"
    assert datasets.load_named("ladybug-1723", 0.01)[1] == "synthetic"


def test_g2o_file_in_the_reference_layout_is_used(data_root):
    g = pkg.synthetic.make_sphere(4, 6)
    folder = data_root / "data" / "odometry" / "3d"
    folder.mkdir(parents=True)
    write_g2o(folder / "sphere2500.g2o", g)
    d, kind, src = datasets.load_pose_graph("sphere2500")
    assert kind == "real" and src == str(folder / "sphere2500.g2o")
    assert d.n_v == g.n_v and d.n_e == g.n_e
    assert np.allclose(d.poses, g.poses, rtol=0, atol=1e-15) and np.allclose(d.meas, g.meas, rtol=0, atol=1e-15)
    (folder / "sphere2500.g2o").unlink()
    d2, kind2, _ = datasets.load_pose_graph("sphere2500", 5, 5)
    assert kind2 == "synthetic" and d2.n_v == 25
