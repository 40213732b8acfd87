"""Per-rank stage times of the distributed solve, measured in a single-GPU lockstep run (the ranks run one after the
other on the same GPU, so each rank's time is what it would take on its own GPU; exchanges are not timed)."""
import sys
sys.path.insert(0, '.')
import numpy as np
import apex_solver_amd as pkg
from apex_solver_amd.solver import GpuSchurComplementSolver, OptimizationType, Problem
name = sys.argv[1] if len(sys.argv) > 1 else 'final-13682'
worlds = [int(w) for w in sys.argv[2].split(',')] if len(sys.argv) > 2 else [2, 4, 8]
d = pkg.synthetic.make_named(name)
prob = Problem.bundle_adjustment(d, OptimizationType.SelfCalibration, 1.0)
for world in worlds:
    ranks = []
    for r in range(world):
        s = GpuSchurComplementSolver(0).with_shard(r, world)
        s.initialize_structure(prob); s.set_parameters(d.poses, d.intr, d.points)
        s.enable_stage_timing(True)
        ranks.append(s)
    GpuSchurComplementSolver.lockstep_solve(ranks, 1e-3)
    for s in ranks: s.reset_stage_times()
    reps = 3
    for _ in range(reps): GpuSchurComplementSolver.lockstep_solve(ranks, 1e-3)
    i0 = ranks[0].info()
    print(name, "world", world, "top columns", i0["dist_top_columns"], "groups", i0["etree_levels"], flush=True)
    for r, s in enumerate(ranks):
        st = s.stage_times(); inf = s.info()
        ms = {k: v[0] / reps for k, v in st.items()}
        print("  rank", r, "share %.3f" % inf["dist_local_fraction"], "obs %.3f" % (inf["local_obs"] / d.n_obs),
              "assemble %.2f" % (ms["cam_reduce"] + ms["landmark_reduce"] + ms["schur_scatter"]),
              "(clears+cam_reduce %.2f landmark_reduce %.2f pairs %.2f)" % (ms["cam_reduce"], ms["landmark_reduce"], ms["schur_scatter"]),
              "factor local %.2f" % ms["factor"], "factor top %.2f" % ms["all_reduce"],
              "tri %.2f" % ms["tri_solve"], "back-subst %.2f" % ms.get("back_substitute", 0.0), flush=True)
    for s in ranks: s.close()
