"""Timing ablations of k_schur_rows (results are WRONG for dbg != 0; timing only).
bits: 1 no LDS atomics, 2 no partner linearisation, 4 no pair phase; dbg >> 8 = KB of extra dynamic LDS."""
import sys, time
sys.path.insert(0, '.')
import apex_solver_amd as pkg
from apex_solver_amd.solver import GpuSchurComplementSolver, OptimizationType, Problem
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
d = pkg.synthetic.make_named('final-13682', scale)
prob = Problem.bundle_adjustment(d, OptimizationType.SelfCalibration, 1.0)
s = GpuSchurComplementSolver(0).initialize_structure(prob)
s.set_parameters(d.poses, d.intr, d.points)
s.enable_stage_timing(True)
for dbg in (0, 1, 2, 3, 4, 8, (8 + (40 << 8)), (8 + (75 << 8))):
    s.set_option("rows_debug", dbg)
    s.assemble(1e-3); s.reset_stage_times()
    for _ in range(3): s.assemble(1e-3)
    st = s.stage_times()
    print("dbg", dbg & 255, "extra_lds_kb", dbg >> 8, "rows ms", round(st["schur_scatter"][0]/3, 3), "pairs", s.info()["pair_blocks"])
