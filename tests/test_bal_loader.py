"""BAL reader (csrc/bal_io.cpp through apex_solver_amd.bal): the reference's loader tests
(crates/apex-io/src/bal.rs:395-690) transcribed, plus the variable construction of
bin/bundle_adjustment.rs:200-257 and a round trip through the text format.  Host only."""
import numpy as np
import pytest

import apex_solver_amd as pkg
from apex_solver_amd.bal import BalLoader, IoError, reference_columns, write_bal


def _minimal(tmp_path, focal=500.0, obs="0 0 -123.456 456.789", header="1 1 1", cam=None, pts=(1.0, 2.0, 3.0)):
    cam = cam if cam is not None else [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, focal, -0.1, 0.05]
    p = tmp_path / "p.txt"
    p.write_text("\n".join([header, obs] + [repr(float(v)) for v in cam] + [repr(float(v)) for v in pts]) + "\n")
    return p


def test_load_minimal_dataset(tmp_path):
    """bal.rs:437-444, 447-462, 465-474, 477-486"""
    ds = BalLoader.load(_minimal(tmp_path))
    assert ds.cameras.shape == (1, 9) and ds.points.shape == (1, 3) and ds.observations.shape == (1, 2)
    assert np.allclose(ds.cameras[0], [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 500.0, -0.1, 0.05], atol=1e-12)
    assert ds.camera_index[0] == 0 and ds.point_index[0] == 0
    assert abs(ds.observations[0, 0] + 123.456) < 1e-10 and abs(ds.observations[0, 1] - 456.789) < 1e-10
    assert np.allclose(ds.points[0], [1.0, 2.0, 3.0], atol=1e-12)


@pytest.mark.parametrize("focal,expect", [(-100.0, 500.0), (0.0, 500.0), (float("inf"), 500.0), (float("nan"), 500.0),
                                          (1234.5, 1234.5), (1e-9, 1e-9)])
def test_normalize_focal_length(tmp_path, focal, expect):
    """bal.rs:489-520 and :100-114: non-positive / non-finite -> DEFAULT_FOCAL_LENGTH, positive preserved"""
    ds = BalLoader.load(_minimal(tmp_path, cam=[0, 0, 0, 0, 0, 0, focal, 0, 0]))
    assert ds.cameras[0, 6] == pytest.approx(expect) and ds.intrinsics[0, 0] == pytest.approx(expect)


def test_load_nonexistent_file(tmp_path):
    with pytest.raises(IoError) as e:
        BalLoader.load(tmp_path / "missing.txt")
    assert e.value.kind == "Io"


def test_load_empty_file(tmp_path):
    p = tmp_path / "e.txt"; p.write_text("")
    with pytest.raises(IoError) as e:
        BalLoader.load(p)
    assert e.value.kind == "Parse"  # Missing header line


@pytest.mark.parametrize("header,kind", [("1 1", "MissingFields"), ("1 abc 1", "InvalidNumber"), ("bad 1 1", "InvalidNumber"),
                                         ("1 1 -3", "InvalidNumber"), ("1 1 1 1", "MissingFields")])
def test_load_header_errors(tmp_path, header, kind):
    """bal.rs:533-560, 629-660"""
    with pytest.raises(IoError) as e:
        BalLoader.load(_minimal(tmp_path, header=header))
    assert e.value.kind == kind


def test_load_truncated_observations(tmp_path):
    p = tmp_path / "t.txt"; p.write_text("1 1 2\n0 0 1.0 1.0\n")
    with pytest.raises(IoError):
        BalLoader.load(p)


def test_load_truncated_cameras(tmp_path):
    p = tmp_path / "t.txt"; p.write_text("1 1 1\n0 0 1.0 1.0\n0.1\n0.2\n0.3\n")
    with pytest.raises(IoError) as e:
        BalLoader.load(p)
    assert e.value.kind == "Parse"


def test_load_observation_errors(tmp_path):
    """bal.rs:611-626: bad coordinate -> InvalidNumber; wrong field count -> MissingFields"""
    with pytest.raises(IoError) as e:
        BalLoader.load(_minimal(tmp_path, obs="0 0 bad_x 1.0"))
    assert e.value.kind == "InvalidNumber"
    with pytest.raises(IoError) as e:
        BalLoader.load(_minimal(tmp_path, obs="0 0 1.0"))
    assert e.value.kind == "MissingFields"


def test_load_multiple_cameras_and_points_with_blank_lines(tmp_path):
    """bal.rs:579-608; blank lines and surrounding whitespace are skipped (:145-150)"""
    lines = ["2 2 3", "", "0 0 1.0 1.0", "  0 1 2.0 2.0  ", "1 0 3.0 3.0"]
    lines += ["0.0"] * 6 + ["100.0", "0.0", "0.0"] + ["0.0"] * 6 + ["200.0", "0.0", "0.0"] + ["", "1", "2", "3", "4", "5", "6"]
    p = tmp_path / "m.txt"; p.write_text("\n".join(lines))
    ds = BalLoader.load(p)
    assert ds.cameras.shape[0] == 2 and ds.points.shape[0] == 2 and ds.observations.shape[0] == 3
    assert ds.cameras[1, 6] == 200.0 and np.allclose(ds.points[1], [4, 5, 6])


def test_variables_match_bundle_adjustment_bin(tmp_path):
    """bin/bundle_adjustment.rs:200-208, 232-246: axis-angle -> SO3 (identity below 1e-10), SE3 vector
    [t, qw,qx,qy,qz], intrinsics [f,k1,k2]."""
    from scipy.spatial.transform import Rotation

    ds = BalLoader.load(_minimal(tmp_path))
    q = Rotation.from_rotvec([0.1, 0.2, 0.3]).as_quat()  # x,y,z,w
    assert np.allclose(ds.poses[0], [0.4, 0.5, 0.6, q[3], q[0], q[1], q[2]], atol=1e-15)
    assert np.allclose(ds.intrinsics[0], [500.0, -0.1, 0.05])
    ds0 = BalLoader.load(_minimal(tmp_path, cam=[1e-12, 0, 0, 1, 2, 3, 400, 0, 0]))
    assert np.array_equal(ds0.poses[0], [1, 2, 3, 1, 0, 0, 0])


def test_text_round_trip_of_a_synthetic_problem(tmp_path):
    d = pkg.synthetic.make_problem(7, 60, 3, 5, config_id=3)
    p = tmp_path / "rt.txt"
    write_bal(p, d)
    back = BalLoader.load(p).to_problem_data()
    assert np.array_equal(back.cam_idx, d.cam_idx) and np.array_equal(back.pt_idx, d.pt_idx)
    assert np.array_equal(back.obs_uv, d.obs_uv) and np.array_equal(back.points, d.points)
    assert np.allclose(back.intr, d.intr, rtol=0, atol=0)
    # quaternion -> axis-angle -> quaternion: same rotation to rounding (sign may flip)
    dot = np.abs(np.sum(back.poses[:, 3:] * d.poses[:, 3:], axis=1))
    assert np.allclose(dot, 1.0, atol=1e-14) and np.allclose(back.poses[:, :3], d.poses[:, :3], atol=0)
    sub = BalLoader.load(p).to_problem_data(num_points=20)   # --num-points of the reference's CLI
    assert sub.n_pt == 20 and sub.pt_idx.max() < 20 and sub.n_obs == int((d.pt_idx < 20).sum())


def test_reference_columns_match_python_layout():
    for n_cam, n_pt in [(3, 10), (12, 100002), (10001, 7)]:
        lay = pkg.layout.reference_column_layout(n_cam, n_pt)
        ic, pc, tc = reference_columns(n_cam, n_pt)
        assert np.array_equal(ic, lay.intr_col) and np.array_equal(pc, lay.pose_col) and np.array_equal(tc, lay.pt_col)
