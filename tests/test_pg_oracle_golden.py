"""The pose-graph oracle against the committed golden fixtures (tests/golden/pg_*.npz, produced by
tests/golden/make_golden_pg.py where the C oracle and the numpy restatement agree)."""
import os

import numpy as np
import pytest

from oracle import pg_oracle as po

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = ["pg_sphere_8x12", "pg_sphere_10x10_huber"]


def load(name):
    return np.load(os.path.join(HERE, "golden", name + ".npz"))


@pytest.mark.parametrize("name", FIXTURES)
def test_oracle_reproduces_golden_iterations(name):
    g = load(name)
    hub = None if float(g["huber_delta"]) <= 0 else float(g["huber_delta"])
    o = po.PgOracle(g["e_from"], g["e_to"], g["meas"], g["pose_col"], g["fix"], hub, g["poses0"])
    assert abs(o.residuals()[0] - float(g["initial_cost"])) <= 1e-13 * float(g["initial_cost"])
    for it in range(int(g["iters"])):
        o.set_params(g[f"it{it}_poses"])
        c, r, J = o.linearize()
        rc, step, grad = o.solve_augmented(float(g[f"it{it}_lambda"]))
        assert rc == 0
        assert np.allclose(r, g[f"it{it}_r"], rtol=0, atol=1e-13 * max(1.0, np.abs(r).max()))
        assert np.allclose(J, g[f"it{it}_J"], rtol=0, atol=1e-13 * np.abs(J).max())
        assert np.allclose(grad, g[f"it{it}_grad"], rtol=0, atol=1e-12 * np.abs(grad).max())
        assert np.linalg.norm(step - g[f"it{it}_step"]) <= 1e-12 * np.linalg.norm(step)
        o.apply_step(step, 1.0)
        nc = o.residuals()[0]
        assert abs(nc - float(g[f"it{it}_new_cost"])) <= 1e-12 * nc


@pytest.mark.parametrize("name", FIXTURES)
def test_oracle_reproduces_golden_lm_history(name):
    g = load(name)
    hub = None if float(g["huber_delta"]) <= 0 else float(g["huber_delta"])
    o = po.PgOracle(g["e_from"], g["e_to"], g["meas"], g["pose_col"], g["fix"], hub, g["poses0"])
    res = o.lm_optimize(po.lm_config(max_iterations=10), hist_rows=16)
    assert res["status"] == int(g["lm_status"]) and res["iterations"] == int(g["lm_iterations"])
    assert np.allclose(res["history"], g["lm_history"], rtol=1e-9, atol=1e-12)
