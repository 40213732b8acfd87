import sys, time
sys.path.insert(0, '.')
import apex_solver_amd as pkg
from apex_solver_amd.solver import GpuSchurComplementSolver, OptimizationType, Problem
d = pkg.synthetic.make_named('final-13682', 0.25)
prob = Problem.bundle_adjustment(d, OptimizationType.SelfCalibration, 1.0)
s = GpuSchurComplementSolver(0).initialize_structure(prob)
s.set_parameters(d.poses, d.intr, d.points)
s.enable_stage_timing(True)
for dbg in (0, 1, 2, 3, 4):
    s.set_option("rows_debug", dbg)
    s.assemble(1e-3); s.reset_stage_times()
    for _ in range(3): s.assemble(1e-3)
    st = s.stage_times()
    print("dbg", dbg, "rows ms", st["schur_scatter"][0]/3, "lm_reduce", st["landmark_reduce"][0]/3, s.info()["pair_blocks"])
