import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import apex_solver_amd as pkg
from apex_solver_amd.solver import GpuSchurComplementSolver, OptimizationType, Problem

d = pkg.synthetic.make_problem(1500, 30000, 3, 7, config_id=310)
prob = Problem.bundle_adjustment(d, OptimizationType.SelfCalibration, 1.0)
lay = prob.layout
def make(shard=None, opts=()):
    s = GpuSchurComplementSolver(0)
    for k, v in opts: s.with_option(k, v)
    if shard: s.with_shard(*shard)
    s.initialize_structure(prob); s.set_parameters(d.poses, d.intr, d.points)
    return s
nc = lay.cam_dof
lam = 1e-4
os.environ["APEX_DIST_EXTRA_SPLITS"] = "5"
s1 = make()
step1 = s1.solve_augmented_equation(lam)
_, gred = s1.get_schur(want_S=False)
world = 2
ranks = [make((r, world)) for r in range(world)]
GpuSchurComplementSolver.lockstep_solve(ranks, lam)
xs = [r.export_step()[0][:nc] for r in ranks]
x = xs[0]
def percam(v):
    return np.maximum(np.abs(v[lay.pose_col[:, None] + np.arange(6)[None]]).max(axis=1), np.abs(v[lay.intr_col[:, None] + np.arange(3)[None]]).max(axis=1))
Sx, _ = s1.schur_matvec(lam, x, implicit=False)
res = percam(Sx - gred)
dx = percam(x - step1[:nc])
order = np.argsort(-res)
print("residual per camera: top 40 cams", order[:40].tolist())
print("  values", [float(f"{res[c]:.1e}") for c in order[:40]])
print("  quantiles of residual per camera", np.quantile(res, [0.5, 0.9, 0.99, 1.0]))
print("  tiles (cam//16) of worst 60:", sorted(set((order[:60] // 16).tolist())))
print("dx per camera quantiles", np.quantile(dx, [0.5, 0.9, 0.99, 1.0]), "worst tiles", sorted(set((np.argsort(-dx)[:60] // 16).tolist())))
# second solve on the same handles (graphs replay) for reproducibility
GpuSchurComplementSolver.lockstep_solve(ranks, lam)
x2 = ranks[0].export_step()[0][:nc]
print("repeat diff", np.linalg.norm(x2 - x) / np.linalg.norm(x))
