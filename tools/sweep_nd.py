import sys, time
sys.path.insert(0, '.')
import apex_solver_amd as pkg
from apex_solver_amd.solver import GpuSchurComplementSolver, OptimizationType, Problem
d = pkg.synthetic.make_named(sys.argv[1] if len(sys.argv) > 1 else 'final-13682')
prob = Problem.bundle_adjustment(d, OptimizationType.SelfCalibration, 1.0)
for leaf in (8, 12, 16, 24, 32, 48):
    s = GpuSchurComplementSolver(0).with_option("nested_dissection", leaf if leaf != 0 else 0)
    t0 = time.time(); s.initialize_structure(prob); ts = time.time() - t0
    s.set_parameters(d.poses, d.intr, d.points)
    s.enable_stage_timing(True)
    s.solve_augmented_equation(1e-3, want_step=False); s.reset_stage_times()
    for _ in range(3): s.solve_augmented_equation(1e-3, want_step=False)
    st = s.stage_times(); inf = s.info()
    print("leaf", leaf, "levels", inf["etree_levels"], "tiles", inf["tiles"], "gemms", inf["n_trsm"] + inf["n_update"], "factor ms", st["factor"][0]/3, "tri ms", st["tri_solve"][0]/3, "setup s", round(ts, 2))
    s.close()
