"""Factor stage of final-13682 by where the dataflow launch starts ("factor_flow" = max columns per level group inside it; -1:
the cost model), with whole-tile update units on / off ("factor_flow_tile").  usage: python tools/flow_tile_sweep.py [workload]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import apex_solver_amd as pkg
from apex_solver_amd.solver import GpuSchurComplementSolver, OptimizationType, Problem
name = sys.argv[1] if len(sys.argv) > 1 else "final-13682"
d = pkg.datasets.load_named(name, 1.0)[0]
prob = Problem.bundle_adjustment(d, OptimizationType.SelfCalibration, 1.0)
for tile in (1, 0):
    for ff in (-1, 8, 13, 17, 32, 64):
        s = GpuSchurComplementSolver(0).with_option("factor_flow_tile", tile).with_option("factor_flow", ff).with_option("factor_flow_rows", 1000)
        t0 = time.perf_counter(); s.initialize_structure(prob); t1 = time.perf_counter()
        s.set_parameters(d.poses, d.intr, d.points)
        for _ in range(3): s.solve_augmented_equation(1e-3, want_step=False)
        s.enable_stage_timing(True); s.reset_stage_times()
        for _ in range(8): s.solve_augmented_equation(1e-3, want_step=False)
        st = s.stage_times()
        print(f"{name} factor_flow_tile={tile} factor_flow={ff:3d}: factor {st['factor'][0] / st['factor'][1]:.3f} ms  groups in the launch {s.counters()['factor_flow_groups']} of {s.info()['etree_levels']}"
              f"  tile plan {s.setup_times()['tile_plan']:.3f} s  set_structure {t1 - t0:.3f} s", flush=True)
        s.close()
