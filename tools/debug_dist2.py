import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import apex_solver_amd as pkg
from apex_solver_amd.solver import GpuSchurComplementSolver, OptimizationType, Problem

d = pkg.synthetic.make_problem(1500, 30000, 3, 7, config_id=310)
prob = Problem.bundle_adjustment(d, OptimizationType.SelfCalibration, 1.0)
def make(shard=None, opts=()):
    s = GpuSchurComplementSolver(0)
    for k, v in opts: s.with_option(k, v)
    if shard: s.with_shard(*shard)
    s.initialize_structure(prob); s.set_parameters(d.poses, d.intr, d.points)
    return s
s1 = make()
nc = prob.layout.cam_dof
lay = prob.layout
for lam in (1e-6, 1e-3, 1.0, 1e3):
    step1 = s1.solve_augmented_equation(lam)
    _, gred = s1.get_schur(want_S=False)
    for world in (2, 3):
        ranks = [make((r, world)) for r in range(world)]
        GpuSchurComplementSolver.lockstep_solve(ranks, lam)
        x = ranks[0].export_step()[0][:nc]
        Sx, _ = s1.schur_matvec(lam, x, implicit=False)
        res = Sx - gred
        ri = np.abs(res[lay.intr_col[:, None] + np.arange(3)[None]]).max(axis=1)
        rp = np.abs(res[lay.pose_col[:, None] + np.arange(6)[None]]).max(axis=1)
        xi = np.abs(x[lay.intr_col[:, None] + np.arange(3)[None]]).max(axis=1)
        worst = np.argsort(-np.maximum(ri, rp))[:6]
        print("lam", lam, "world", world, "resid", np.linalg.norm(res) / np.linalg.norm(gred), "|g|", np.linalg.norm(gred), "|x|", np.linalg.norm(x),
              "worst", [(int(c), float(f"{rp[c]:.2e}"), float(f"{ri[c]:.2e}")) for c in worst])
        # ratio residual / (lam * x) on the worst rows
        c = worst[0]
        cols = np.concatenate([lay.pose_col[c] + np.arange(6), lay.intr_col[c] + np.arange(3)])
        print("     res/x on worst camera", res[cols] / x[cols])
        for s in ranks: s.close()
