# tools/sched_probe.py '<json opts>' : one solve with the given implementation switches (fresh process per set)
import sys, os, json; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, apex_solver_amd as pkg
from apex_solver_amd import datasets
from apex_solver_amd.solver import GpuSchurComplementSolver, OptimizationType, Problem
opts = json.loads(sys.argv[1]) if len(sys.argv) > 1 else {}
d,_,_ = datasets.load_named("final-13682", float(os.environ.get("PROBE_SCALE", "0.1")))
prob = Problem.bundle_adjustment(d, OptimizationType.SelfCalibration, 1.0)
s = GpuSchurComplementSolver(0)
for k, v in opts.items(): s.with_option(k, v)
try:
    s.initialize_structure(prob); s.set_parameters(d.poses, d.intr, d.points)
    a = s.solve_augmented_equation(1e-3); b = s.solve_augmented_equation(1e-3)
    print(opts, "ok", np.linalg.norm(a), np.linalg.norm(a-b)/np.linalg.norm(a), s.info()["etree_levels"])
except Exception as e:
    print(opts, "FAILED", str(e)[:120])
