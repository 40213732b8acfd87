"""Where the set-up of one optimize() goes on the headline shape: apexgpu_set_structure by sub-phase (APEX_SETUP_TRACE=1 on
stderr), the parameter upload right behind it, the first cost and the first two solves.  APEX_SETUP_FREE=sync|leak changes how
the host lists are released (default: a background thread)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import apex_solver_amd as pkg
from apex_solver_amd.solver import GpuSchurComplementSolver, OptimizationType, Problem
name = sys.argv[1] if len(sys.argv) > 1 else "final-13682"
d = pkg.datasets.load_named(name, 1.0)[0]
prob = Problem.bundle_adjustment(d, OptimizationType.SelfCalibration, 1.0)
T = time.perf_counter
for rep in range(int(os.environ.get("REPS", "2"))):
    t0 = T(); s = GpuSchurComplementSolver(0).initialize_structure(prob); t1 = T()
    s.set_parameters(d.poses, d.intr, d.points); t2 = T()
    c = s.compute_cost(); t3 = T()
    s.solve_augmented_equation(1e-3, want_step=False); t4 = T()
    s.solve_augmented_equation(1e-3, want_step=False); t5 = T()
    print(f"rep {rep} free={os.environ.get('APEX_SETUP_FREE', 'bg')}: initialize_structure {t1 - t0:.3f}  set_parameters {t2 - t1:.3f}  first cost {t3 - t2:.3f}  "
          f"first solve {t4 - t3:.3f}  second solve {t5 - t4:.3f}  | phases {({k: round(v, 3) for k, v in s.setup_times().items() if isinstance(v, float)})} wall {s.setup_wall}", flush=True)
    s.close()
