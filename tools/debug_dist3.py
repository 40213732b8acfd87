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
for world in (2, 3):
    for extra in (0, 1, 2, 3, 5):
        os.environ["APEX_DIST_EXTRA_SPLITS"] = str(extra)
        ranks = [make((r, world)) for r in range(world)]
        GpuSchurComplementSolver.lockstep_solve(ranks, lam)
        x = ranks[0].export_step()[0][:nc]
        Sx, _ = s1.schur_matvec(lam, x, implicit=False)
        inf = [s.info() for s in ranks]
        print("world", world, "extra", extra, "top", inf[0]["dist_top_columns"], "frac", [round(i["dist_local_fraction"], 3) for i in inf],
              "resid", np.linalg.norm(Sx - gred) / np.linalg.norm(gred))
        for s in ranks: s.close()
