"""PriorFactor blocks of the pose-graph path (src/factors/prior_factor.rs:96-108; the gauge of the reference's pose-graph
integration test, tests/integration_tests.rs:98-118: a PriorFactor with Huber(1.0) on the first vertex instead of a
fixed variable).  Known answers from the reference's own doc tests, then the oracle (oracle/pg_oracle.c) against an
independent dense numpy restatement of the whole system -- residual vector [prior (7); edges (6 each)], Jacobian with
the prior's 7 x 7 identity truncated to the variable's six tangent columns (src/linearizer/cpu/sparse.rs:201-204)."""
import numpy as np
import pytest

import apex_solver_amd as pkg
from apex_solver_amd.pose_graph import PoseGraphProblem, se3_as_vector
from oracle import pg_oracle as po


def rel(a, b):
    return float(np.linalg.norm(np.ravel(a) - np.ravel(b)) / max(np.linalg.norm(np.ravel(b)), 1e-300))


def huber_scale(delta, s):
    """sqrt(rho'(s)) of HuberLoss (loss_functions.rs:364-380) as the Corrector applies it (corrector.rs:143-181)."""
    if delta is None or delta <= 0 or s <= delta * delta:
        return 1.0
    return np.sqrt(delta / np.sqrt(s))


def dense_system(o, prob):
    """[r; J] of the whole problem in the reference's row order (the prior blocks were added first), dense."""
    c, r, J = o.linearize()
    n = 6 * o.n_v
    d = prob.data
    rows_r, rows_J = [], []
    poses = o.get_params()
    for (v, x, dl) in prob.priors:
        rp = se3_as_vector(poses[v]) - x
        sc = huber_scale(dl, float(rp @ rp))
        Jp = np.zeros((7, n))
        Jp[:6, prob.pose_col[v]:prob.pose_col[v] + 6] = np.eye(6)
        rows_r.append(sc * rp); rows_J.append(sc * Jp)
    for e in range(o.n_e):
        Je = np.zeros((6, n))
        Je[:, prob.pose_col[d.e_from[e]]:prob.pose_col[d.e_from[e]] + 6] += J[e][:, :6]
        Je[:, prob.pose_col[d.e_to[e]]:prob.pose_col[d.e_to[e]] + 6] += J[e][:, 6:]
        rows_r.append(r[e]); rows_J.append(Je)
    return c, np.concatenate(rows_r), np.vstack(rows_J)


def test_prior_factor_doc_examples():
    """prior_factor.rs:72-91: residual = x - data, Jacobian = identity -- on an SE3 variable through the oracle."""
    d = pkg.synthetic.make_sphere(3, 4, config_id=5)
    prob = PoseGraphProblem(d).add_prior(f"x{int(d.ids[0])}", data=se3_as_vector(d.poses[0]) - np.array([0.5, 0.3, 0, 0, 0, 0, 0]))
    o = po.PgOracle.from_problem(prob)
    o.residuals()
    rp = o.prior_residuals()
    assert rp.shape == (1, 7) and abs(rp[0, 0] - 0.5) < 1e-10 and abs(rp[0, 1] - 0.3) < 1e-10 and np.abs(rp[0, 2:]).max() < 1e-15
    assert prob.num_residual_blocks == d.n_e + 1


@pytest.mark.parametrize("delta", [None, 1.0, 0.05])
def test_oracle_with_priors_matches_dense_restatement(delta):
    d = pkg.synthetic.make_sphere(4, 6, config_id=7)
    rng = np.random.default_rng(3)
    prob = PoseGraphProblem(d, huber_delta=0.7)            # no fixed DOF: the priors are the gauge
    x0 = se3_as_vector(d.poses[0]); x0[:3] += 0.2 * rng.standard_normal(3); x0[3:] += 0.05 * rng.standard_normal(4)
    prob.add_prior(f"x{int(d.ids[0])}", data=x0, huber_delta=delta)
    prob.add_prior(f"x{int(d.ids[5])}", huber_delta=delta)          # data = the initial value: zero residual
    o = po.PgOracle.from_problem(prob)
    c, r, J = dense_system(o, prob)
    assert abs(c - 0.5 * float(r @ r)) <= 1e-13 * c
    if delta == 0.05:
        assert huber_scale(delta, float(np.sum((se3_as_vector(d.poses[0]) - x0) ** 2))) < 1.0   # the loss really acts
    H, g = o.normal_equations()
    assert rel(H, J.T @ J) < 1e-13 and rel(g, J.T @ r) < 1e-13
    assert rel(o.column_norms(), np.sqrt(np.sum(J * J, axis=0))) < 1e-13
    lam = 1e-4
    rc, step, grad = o.solve_augmented(lam)
    assert rc == 0
    x = np.linalg.solve(J.T @ J + lam * np.eye(J.shape[1]), -(J.T @ r))
    assert rel(step, x) < 1e-9 and rel(grad, J.T @ r) < 1e-13
    # residual-only evaluation agrees with the linearisation's cost; the gauge is held without a fixed variable
    assert abs(o.residuals()[0] - c) <= 1e-14 * c
    assert po.PgOracle.from_problem(PoseGraphProblem(d, huber_delta=0.7)).__class__ is po.PgOracle
    free = po.PgOracle.from_problem(PoseGraphProblem(d, huber_delta=0.7)); free.linearize()
    assert free.solve_augmented(0.0)[0] != 0            # the same graph without priors and without damping is singular ...
    o.set_column_scaling(None); o.linearize()
    assert o.solve_augmented(0.0)[0] == 0               # ... with the priors it is not
    # Jacobi scaling sees the prior's columns too
    sc = 1.0 / (1.0 + o.column_norms())
    o.set_column_scaling(sc)
    rc, y, gs = o.solve_augmented(lam)
    Js = J * sc[None, :]
    ys = np.linalg.solve(Js.T @ Js + lam * np.eye(J.shape[1]), -(Js.T @ r))
    assert rc == 0 and rel(y, ys) < 1e-9 and rel(gs, Js.T @ r) < 1e-13


def test_lm_with_prior_gauge_like_the_integration_test():
    """tests/integration_tests.rs:98-118, 150-190: prior (Huber 1.0) on the first vertex, nothing fixed; the run must
    converge and reduce the cost -- the reference's own assertions -- and the first vertex must stay near its prior."""
    d = pkg.synthetic.make_sphere(6, 8, config_id=9)
    prob = PoseGraphProblem(d).add_prior(f"x{int(d.ids[0])}", huber_delta=1.0)
    o = po.PgOracle.from_problem(prob)
    cfg = po.lm_config(max_iterations=50)
    out = o.lm_optimize(cfg)
    c0, c1 = out["initial_cost"], out["final_cost"]
    assert out["iterations"] >= 2 and c1 < c0 * 0.5
    # (a soft constraint of unit weight against the edges' pull: the vertex stays near its prior, not on it)
    assert np.abs(se3_as_vector(o.get_params()[0]) - se3_as_vector(d.poses[0])).max() < 0.1
