"""Where the 0.18 s of the first apexgpu_set_params go: the call twice in a row (the second has its pinned chunks and pays
only for the copy), on the headline shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("APEX_SYNTH_CACHE", "/tmp/apex_synth_cache")
import numpy as np
import apex_solver_amd as pkg
from apex_solver_amd.solver import GpuSchurComplementSolver, OptimizationType, Problem
d = pkg.datasets.load_named("final-13682", 1.0)[0]
prob = Problem.bundle_adjustment(d, OptimizationType.SelfCalibration, 1.0)
s = GpuSchurComplementSolver(0).initialize_structure(prob)
time.sleep(1.0)
for rep in range(3):
    t = time.perf_counter(); s.set_parameters(d.poses, d.intr, d.points); print("set_parameters call %d: %.1f ms" % (rep, (time.perf_counter() - t) * 1e3))
t = time.perf_counter(); c = s.compute_cost(); print("first cost: %.1f ms" % ((time.perf_counter() - t) * 1e3))
s.close()
