"""Generate the committed golden fixtures (tests/golden/*.npz).

The Rust reference cannot be executed in this environment (no cargo/rustc), so these
vectors are NOT reference outputs.  They are outputs on which two independent
restatements of the reference algorithm agree -- the C oracle (oracle/ba_oracle.c)
and the numpy/scipy restatement (tests/np_ref.py, direct solve of the full damped
normal equations) -- and the script refuses to write a fixture where they disagree.
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import apex_solver_amd as pkg  # noqa: E402
import np_ref  # noqa: E402
from oracle import oracle as ora  # noqa: E402


def rel(a, b):
    return float(np.linalg.norm(np.ravel(a) - np.ravel(b)) / max(np.linalg.norm(np.ravel(b)), 1e-300))


def make(name, n_cam, n_pt, k_lo, k_hi, cid, mode, behind=0.0, iters=3):
    d = pkg.synthetic.make_problem(n_cam, n_pt, k_lo, k_hi, config_id=cid, behind_frac=behind)
    lay = pkg.layout.reference_column_layout(n_cam, n_pt)
    p = ora.from_data(d, lay, mode=mode)
    out = dict(
        n_cam=n_cam, n_pt=n_pt, mode=mode, huber_delta=1.0,
        poses0=d.poses, intr0=d.intr, points0=d.points, cam_idx=d.cam_idx, pt_idx=d.pt_idx, obs_uv=d.obs_uv,
        intr_col=lay.intr_col, pose_col=lay.pose_col, pt_col=lay.pt_col,
    )
    lam, nu = 1e-3, 2.0
    cost = p.residuals()[0]
    out["initial_cost"] = cost
    for it in range(iters):
        poses, intr, pts = p.get_params()
        c, r, Jp, Jl, Ji = p.linearize()
        step, grad, S, gred = p.solve_augmented(lam, 0, want_schur=True)
        # --- independent check ------------------------------------------------
        rt, c2, Jp2, Jl2, Ji2 = np_ref.jacobian_blocks(poses, intr, pts, d.cam_idx, d.pt_idx, d.obs_uv)
        J = np_ref.sparse_jacobian(Jp2, Jl2, Ji2, d.cam_idx, d.pt_idx, lay, selfcal=(mode == "selfcal"))
        dx, g, H = np_ref.direct_step(J, rt.ravel(), lam)
        S2, gred2 = np_ref.schur_dense(H, g, lay.cam_dof, lam)
        chk = dict(r=rel(r, rt.ravel()), Jp=rel(Jp, Jp2), Jl=rel(Jl, Jl2), Ji=rel(Ji, Ji2), grad=rel(grad, g),
                   S=rel(S, S2), gred=rel(gred, gred2), step=rel(step, dx), cost=abs(c - c2) / c2)
        print(name, "iter", it, {k: f"{v:.1e}" for k, v in chk.items()})
        assert max(chk[k] for k in ("r", "Jp", "Jl", "Ji", "grad", "S", "gred", "cost")) < 1e-11, chk
        assert chk["step"] < 1e-6, chk  # Schur vs direct solve of an ill-conditioned system
        # --- referee: the exact step of this linearisation (oracle/ba_oracle.c, ora_solve_augmented_quad) and a
        # WELL-CONDITIONED solve of the same linearisation (lambda = 1e4: cond(S) ~ 1e2..1e3), where fp64 solvers must
        # agree to the north star's 1e-10 outright
        exact, info = p.solve_augmented_quad(lam)
        wc_lam = 1e4
        wstep, _, wS, wgred = p.solve_augmented(wc_lam, 0, want_schur=True)
        wexact, winfo = p.solve_augmented_quad(wc_lam)
        wdx = np_ref.direct_step(J, rt.ravel(), wc_lam)[0]
        cond_wc = float(np.linalg.cond(wS))
        print(name, "iter", it, "fp64 oracle vs exact", f"{rel(step, exact):.1e}", info, "| lambda 1e4: cond(S)", f"{cond_wc:.1e}",
              "oracle vs exact", f"{rel(wstep, wexact):.1e}", "vs numpy direct", f"{rel(wstep, wdx):.1e}")
        assert info["residual"] < 1e-28 and winfo["residual"] < 1e-28
        assert cond_wc < 1e5 and rel(wstep, wexact) < 1e-13 and rel(wstep, wdx) < 1e-11
        out.update({f"it{it}_step_exact": exact, f"it{it}_wc_lambda": wc_lam, f"it{it}_wc_step": wstep,
                    f"it{it}_wc_step_exact": wexact, f"it{it}_wc_S": wS, f"it{it}_wc_gred": wgred})
        pred = 0.5 * float(step @ (lam * step - grad))
        p.apply_step(step, 1.0)
        new_cost = p.residuals()[0]
        # trial point cost against numpy's retraction
        fixp = np.zeros((n_cam, 6), dtype=bool); fixp[0] = True
        tp, ti, tl = np_ref.retract(poses, intr, pts, step, lay, fix_pose=fixp)
        nc2 = np_ref.residuals(tp, ti, tl, d.cam_idx, d.pt_idx, d.obs_uv)[1]
        assert abs(new_cost - nc2) / nc2 < 1e-11, (new_cost, nc2)
        rho = (cost - new_cost) / pred
        out.update({f"it{it}_poses": poses, f"it{it}_intr": intr, f"it{it}_points": pts})
        out.update({f"it{it}_lambda": lam, f"it{it}_cost": c, f"it{it}_r": r, f"it{it}_Jpose": Jp, f"it{it}_Jpt": Jl,
                    f"it{it}_Jintr": Ji, f"it{it}_grad": grad, f"it{it}_S": S, f"it{it}_gred": gred,
                    f"it{it}_step": step, f"it{it}_pred": pred, f"it{it}_new_cost": new_cost, f"it{it}_rho": rho})
        if rho > 0:
            coff = 2 * rho - 1
            lam = max(lam * max(1 / 3, 1 - coff**3), 1e-12); nu = 2.0; cost = new_cost
        else:
            lam = min(lam * nu, 1e12); nu *= 2; p.apply_step(step, -1.0)
        out[f"it{it}_accepted"] = bool(rho > 0)
    poses, intr, pts = p.get_params()
    out.update(poses_end=poses, intr_end=intr, points_end=pts, iters=iters)
    # whole-loop history with the reference's default LM config
    p2 = ora.from_data(d, lay, mode=mode)
    res = p2.optimize(ora.LMConfig.default(max_iterations=8))
    out.update(lm_history=res.history, lm_status=res.status, lm_iterations=res.iterations)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


def augment(name):
    """Add the referee vectors (it*_step_exact, it*_wc_*) to a fixture that predates them WITHOUT touching what it already
    holds: the iteration's stored parameters are re-linearised (the oracle must reproduce the stored r / J / step bit for
    bit, else this refuses) and the two referee solves are appended."""
    path = os.path.join(HERE, name + ".npz")
    g = dict(np.load(path))
    lay = pkg.layout.reference_column_layout(int(g["n_cam"]), int(g["n_pt"]))
    d = pkg.synthetic.BAProblemData(poses=g["poses0"], intr=g["intr0"], points=g["points0"], cam_idx=g["cam_idx"],
                                    pt_idx=g["pt_idx"], obs_uv=g["obs_uv"], name=name)
    mode = str(g["mode"])
    p = ora.from_data(d, lay, mode=mode)
    for it in range(int(g["iters"])):
        p.set_params(g[f"it{it}_poses"], g[f"it{it}_intr"], g[f"it{it}_points"])
        c, r, Jp, Jl, Ji = p.linearize()
        lam = float(g[f"it{it}_lambda"])
        step, grad, S, gred = p.solve_augmented(lam, 0, want_schur=True)
        assert np.array_equal(r, g[f"it{it}_r"]) and np.array_equal(Jp, g[f"it{it}_Jpose"]) and np.array_equal(step, g[f"it{it}_step"]), \
            "the oracle no longer reproduces the stored fixture bit for bit"
        exact, info = p.solve_augmented_quad(lam)
        wc_lam = 1e4
        wstep, _, wS, wgred = p.solve_augmented(wc_lam, 0, want_schur=True)
        wexact, winfo = p.solve_augmented_quad(wc_lam)
        rt, c2, Jp2, Jl2, Ji2 = np_ref.jacobian_blocks(g[f"it{it}_poses"], g[f"it{it}_intr"], g[f"it{it}_points"], d.cam_idx, d.pt_idx, d.obs_uv)
        J = np_ref.sparse_jacobian(Jp2, Jl2, Ji2, d.cam_idx, d.pt_idx, lay, selfcal=(mode == "selfcal"))
        wdx = np_ref.direct_step(J, rt.ravel(), wc_lam)[0]
        cond_wc = float(np.linalg.cond(wS))
        print(name, "iter", it, "fp64 oracle vs exact", f"{rel(step, exact):.1e}", info, "| lambda 1e4: cond(S)", f"{cond_wc:.1e}",
              "oracle vs exact", f"{rel(wstep, wexact):.1e}", "vs numpy direct", f"{rel(wstep, wdx):.1e}")
        assert info["residual"] < 1e-28 and winfo["residual"] < 1e-28
        assert cond_wc < 1e5 and rel(wstep, wexact) < 1e-13 and rel(wstep, wdx) < 1e-11
        g.update({f"it{it}_step_exact": exact, f"it{it}_wc_lambda": wc_lam, f"it{it}_wc_step": wstep,
                  f"it{it}_wc_step_exact": wexact, f"it{it}_wc_S": wS, f"it{it}_wc_gred": wgred})
    np.savez_compressed(path, **g)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "augment":   # round 3: referee vectors added to the round-1 fixtures in place
        for n in ("ba6x40_selfcal", "ba6x40_ba", "ba9x120_selfcal_behind"):
            augment(n)
        sys.exit(0)
    make("ba6x40_selfcal", 6, 40, 3, 5, 101, "selfcal")
    make("ba6x40_ba", 6, 40, 3, 5, 101, "ba")
    make("ba9x120_selfcal_behind", 9, 120, 3, 6, 102, "selfcal", behind=0.05)
