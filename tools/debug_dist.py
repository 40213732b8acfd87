import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import apex_solver_amd as pkg
from apex_solver_amd.solver import GpuSchurComplementSolver, OptimizationType, Problem

d = pkg.synthetic.make_problem(1500, 30000, 3, 7, config_id=310)
lam = 1e-3
prob = Problem.bundle_adjustment(d, OptimizationType.SelfCalibration, 1.0)
def make(shard=None, opts=()):
    s = GpuSchurComplementSolver(0)
    for k, v in opts: s.with_option(k, v)
    if shard: s.with_shard(*shard)
    s.initialize_structure(prob); s.set_parameters(d.poses, d.intr, d.points)
    return s
s1 = make()
step1 = s1.solve_augmented_equation(lam)
_, gred = s1.get_schur(want_S=False)
nc = prob.layout.cam_dof
for world in (2, 3, 4):
    for opts in ((), (("graphs", 0),), (("update_overlap", 0),), (("graphs", 0), ("update_overlap", 0))):
        ranks = [make((r, world), opts) for r in range(world)]
        GpuSchurComplementSolver.lockstep_solve(ranks, lam)
        x = ranks[0].export_step()[0][:nc]
        Sx, _ = s1.schur_matvec(lam, x, implicit=False)
        res = Sx - gred
        print("world", world, opts, "resid", np.linalg.norm(res) / np.linalg.norm(gred), "stepdiff", np.linalg.norm(x - step1[:nc]) / np.linalg.norm(step1[:nc]))
        if opts == ():
            # where is the residual? by camera (pose cols)
            lay = prob.layout
            rc = np.abs(res[lay.pose_col[:, None] + np.arange(6)[None]]).max(axis=1)
            top = np.argsort(-rc)[:12]
            print("   worst cameras", sorted(top.tolist()), "max", rc.max(), "median", np.median(rc))
        for s in ranks: s.close()
