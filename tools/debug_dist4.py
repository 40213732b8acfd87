import sys, os
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
lam = 1e-4
step1 = s1.solve_augmented_equation(lam)
_, gred = s1.get_schur(want_S=False)
for world in (2, 3, 4):
    for extra in (0, 3, 5):
        os.environ["APEX_DIST_EXTRA_SPLITS"] = str(extra)
        os.environ["APEX_DIST_SELFTEST"] = str(world)
        s = make()
        x = s.solve_augmented_equation(lam)[:nc]
        Sx, _ = s1.schur_matvec(lam, x, implicit=False)
        inf = s.info()
        print("selftest world", world, "extra", extra, "top", inf["dist_top_columns"], "levels", inf["etree_levels"],
              "resid", np.linalg.norm(Sx - gred) / np.linalg.norm(gred))
        s.close()
