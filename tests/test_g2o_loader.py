"""G2O reader (apex-solver_amd/csrc/g2o_io.cpp through pose_graph.G2oLoader) against the assertions of
the reference's own loader tests (crates/apex-io/src/g2o.rs:625-1460).  Host only."""
import numpy as np
import pytest

import apex_solver_amd as pkg
from apex_solver_amd.pose_graph import G2oError, G2oLoader, pose_graph_columns, write_g2o

INFO = "100.0 0.0 0.0 0.0 0.0 0.0 100.0 0.0 0.0 0.0 0.0 100.0 0.0 0.0 0.0 100.0 0.0 0.0 100.0 0.0 100.0"


def load_text(tmp_path, text):
    p = tmp_path / "g.g2o"
    p.write_text(text)
    return G2oLoader.load(p)


def test_parse_vertex_se3(tmp_path):
    g = load_text(tmp_path, "VERTEX_SE3:QUAT 1 1.0 2.0 3.0 0.0 0.0 0.0 1.0\n")
    assert g.vertex_ids.tolist() == [1]
    assert np.array_equal(g.vertex_poses[0, :3], [1.0, 2.0, 3.0])
    assert g.vertex_poses[0, 3] > 0.99


def test_edge_information_matrix(tmp_path):
    g = load_text(tmp_path, "VERTEX_SE3:QUAT 0 0.0 0.0 0.0 0.0 0.0 0.0 1.0\nVERTEX_SE3:QUAT 1 1.0 0.0 0.0 0.0 0.0 0.0 1.0\n"
                  f"EDGE_SE3:QUAT 0 1 1.0 0.0 0.0 0.0 0.0 0.0 1.0 {INFO}\n")
    assert g.edge_from.tolist() == [0] and g.edge_to.tolist() == [1]
    assert abs(g.edge_information[0, 0, 0] - 100.0) < 1e-10 and abs(g.edge_information[0, 1, 1] - 100.0) < 1e-10
    assert np.array_equal(g.edge_information[0], g.edge_information[0].T)
    assert np.allclose(g.edge_measurements[0], [1, 0, 0, 1, 0, 0, 0])


@pytest.mark.parametrize("text,kind", [
    ("VERTEX_SE3:QUAT 0 0.0 0.0 0.0 0.0 0.0 0.0 0.1\n", "InvalidQuaternion"),
    ("VERTEX_SE3:QUAT 0 1.0 2.0\n", "MissingFields"),
    ("VERTEX_SE3:QUAT 0 bad 0.0 0.0 0.0 0.0 0.0 1.0\n", "InvalidNumber"),
    ("VERTEX_SE3:QUAT 0 0.0 0.0 0.0 0.0 bad 0.0 1.0\n", "InvalidNumber"),
    ("VERTEX_SE3:QUAT x 0.0 0.0 0.0 0.0 0.0 0.0 1.0\n", "InvalidNumber"),
    ("VERTEX_SE3:QUAT -1 0.0 0.0 0.0 0.0 0.0 0.0 1.0\n", "InvalidNumber"),
    ("VERTEX_SE3:QUAT 0 0 0 0 0 0 0 1\nVERTEX_SE3:QUAT 0 1 0 0 0 0 0 1\n", "DuplicateVertex"),
    ("EDGE_SE3:QUAT 0 1 1.0 0.0\n", "MissingFields"),
    (f"EDGE_SE3:QUAT 0 1 bad 0.0 0.0 0.0 0.0 0.0 1.0 {INFO}\n", "InvalidNumber"),
    (f"EDGE_SE3:QUAT 0 1 1.0 0.0 0.0 0.0 bad 0.0 1.0 {INFO}\n", "InvalidNumber"),
    (f"EDGE_SE3:QUAT a 1 1.0 0.0 0.0 0.0 0.0 0.0 1.0 {INFO}\n", "InvalidNumber"),
    (f"EDGE_SE3:QUAT 0 b 1.0 0.0 0.0 0.0 0.0 0.0 1.0 {INFO}\n", "InvalidNumber"),
    ("EDGE_SE3:QUAT 0 1 1.0 0.0 0.0 0.0 0.0 0.0 1.0 bad" + INFO[5:] + "\n", "Parse"),
    ("VERTEX_SE2 invalid 1.0 2.0 0.5\n", "InvalidNumber"),
    ("VERTEX_SE2 0\n", "MissingFields"),
    ("VERTEX_SE2 0 0 0 0\nVERTEX_SE2 0 1 1 0\n", "DuplicateVertex"),
    ("EDGE_SE2 0 1 1.0\n", "MissingFields"),
    ("EDGE_SE2 0 1 1.0 0.0 0.0 bad 0 0 1 0 1\n", "Parse"),
])
def test_error_kinds(tmp_path, text, kind):
    with pytest.raises(G2oError) as e:
        load_text(tmp_path, text)
    assert e.value.kind == kind


def test_missing_file():
    with pytest.raises(G2oError) as e:
        G2oLoader.load("/nonexistent/path/file.g2o")
    assert e.value.kind == "Io"


def test_comments_blank_lines_and_unknown_tags_are_skipped(tmp_path):
    g = load_text(tmp_path, "# comment\nVERTEX_SE3:QUAT 0 0 0 0 0 0 0 1\n\nFIX 0\n  VERTEX_SE3:QUAT 1 1 0 0 0 0 0 1  \nVERTEX_SE2 7 1 2 0.5\n")
    assert g.vertex_ids.tolist() == [0, 1] and g.n_vertices_se2 == 1
    assert g.vertex_count() == 3


def test_write_round_trip(tmp_path):
    d = pkg.synthetic.make_sphere(6, 8, id_stride=3)
    p = tmp_path / "s.g2o"
    write_g2o(p, d)
    g = G2oLoader.load(p)
    q = g.to_problem_data()
    assert np.array_equal(q.ids, d.ids) and np.array_equal(q.e_from, d.e_from) and np.array_equal(q.e_to, d.e_to)
    assert np.abs(q.poses - d.poses).max() < 1e-15 and np.abs(q.meas - d.meas).max() < 1e-15
    assert np.allclose(g.edge_information[0], np.eye(6))


def test_large_file_and_sorted_problem(tmp_path):
    """> 1000 lines (the reference's parallel path, g2o.rs:1367-1430) with vertices in shuffled order."""
    d = pkg.synthetic.make_sphere(30, 40)
    rng = np.random.default_rng(0)
    perm = rng.permutation(d.n_v)
    shuffled = pkg.synthetic.PoseGraphData(ids=d.ids[perm], poses=d.poses[perm], e_from=np.argsort(perm)[d.e_from].astype(np.uint32),
                                           e_to=np.argsort(perm)[d.e_to].astype(np.uint32), meas=d.meas)
    p = tmp_path / "big.g2o"
    write_g2o(p, shuffled)
    g = G2oLoader.load(p)
    assert g.vertex_ids.shape[0] == 1200 and g.edge_from.shape[0] == d.n_e
    q = g.to_problem_data()
    assert np.array_equal(q.ids, d.ids) and np.array_equal(q.e_from, d.e_from)
    assert np.abs(q.poses - d.poses).max() < 1e-15


def test_reference_column_order():
    """Variables are named x{id} and columns follow the SORTED NAMES (src/optimizer/mod.rs:530-536):
    x0, x1, x10, x11, ..., x2, ..."""
    ids = np.arange(12, dtype=np.int64)
    col = pose_graph_columns(ids)
    names = sorted(f"x{i}" for i in ids)
    assert [int(col[int(n[1:])]) for n in names] == [6 * r for r in range(12)]
    assert col[10] == 12 and col[2] == 24
    prob = pkg.PoseGraphProblem.pose_graph(pkg.synthetic.make_sphere(3, 4))
    assert prob.fix[0].tolist() == [1] * 6 and prob.fix[1:].sum() == 0
    assert prob.total_dof == 72 and prob.num_residual_blocks == 11 + 8
