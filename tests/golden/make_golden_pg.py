"""Generate the committed pose-graph golden fixtures (tests/golden/pg_*.npz).

As for the BA fixtures, the Rust reference cannot be executed here (no cargo/rustc): these are outputs
on which two independent restatements agree -- the C oracle (oracle/pg_oracle.c, quaternion algebra,
envelope Cholesky) and tests/np_ref_pg.py (rotation matrices + scipy, dense numpy solve) -- and the
script refuses to write a fixture where they disagree.  Run:  python tests/golden/make_golden_pg.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import apex_solver_amd as pkg  # noqa: E402
import np_ref_pg  # noqa: E402
from oracle import pg_oracle as po  # noqa: E402


def rel(a, b):
    return float(np.linalg.norm(np.ravel(a) - np.ravel(b)) / max(np.linalg.norm(np.ravel(b)), 1e-300))


def make(name, rings, per_ring, huber, iters=4, id_stride=1):
    d = pkg.synthetic.make_sphere(rings, per_ring, id_stride=id_stride)
    prob = pkg.PoseGraphProblem.pose_graph(d, huber)
    o = po.PgOracle.from_problem(prob)
    out = dict(ids=d.ids, poses0=d.poses, e_from=d.e_from, e_to=d.e_to, meas=d.meas, pose_col=prob.pose_col, fix=prob.fix,
               huber_delta=-1.0 if huber is None else huber)
    lam, nu = 1e-3, 2.0
    cost = o.residuals()[0]
    out["initial_cost"] = cost
    for it in range(iters):
        poses = o.get_params()
        c, r, J = o.linearize()
        rc, step, grad = o.solve_augmented(lam)
        assert rc == 0
        r2, J2 = np_ref_pg.linearize(poses, d.e_from, d.e_to, d.meas, huber)
        H2, g2 = np_ref_pg.normal_equations(r2, J2, d.e_from, d.e_to, prob.pose_col)
        A = H2 + lam * np.eye(H2.shape[0])
        dx = np.linalg.solve(A, -g2)
        chk = dict(r=rel(r, r2), J=rel(J, J2), grad=rel(grad, g2), step=rel(step, dx), cost=abs(c - 0.5 * np.sum(r2 * r2)) / c)
        print(name, "iter", it, {k: f"{v:.1e}" for k, v in chk.items()}, "cond", f"{np.linalg.cond(A):.1e}")
        assert max(chk[k] for k in ("r", "J", "grad", "cost")) < 1e-11, chk
        assert chk["step"] < 1e-7, chk
        pred = 0.5 * float(step @ (lam * step - grad))
        o.apply_step(step, 1.0)
        new_cost = o.residuals()[0]
        rho = (cost - new_cost) / pred
        out.update({f"it{it}_poses": poses, f"it{it}_lambda": lam, f"it{it}_cost": c, f"it{it}_r": r, f"it{it}_J": J,
                    f"it{it}_grad": grad, f"it{it}_step": step, f"it{it}_pred": pred, f"it{it}_new_cost": new_cost,
                    f"it{it}_rho": rho, f"it{it}_accepted": bool(rho > 0), f"it{it}_cond": np.linalg.cond(A)})
        if rho > 0:
            coff = 2 * rho - 1
            lam = max(lam * max(1 / 3, 1 - coff**3), 1e-12); nu = 2.0; cost = new_cost
        else:
            lam = min(lam * nu, 1e12); nu *= 2; o.apply_step(step, -1.0)
    out.update(poses_end=o.get_params(), iters=iters)
    o2 = po.PgOracle.from_problem(prob)
    res = o2.lm_optimize(po.lm_config(max_iterations=10), hist_rows=16)
    out.update(lm_history=res["history"], lm_status=res["status"], lm_iterations=res["iterations"], lm_final_cost=res["final_cost"])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


if __name__ == "__main__":
    make("pg_sphere_8x12", 8, 12, None)
    make("pg_sphere_10x10_huber", 10, 10, 1.0, id_stride=7)
