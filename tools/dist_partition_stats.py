"""Partition statistics of the distributed factorisation on a named shape (no solve): top columns and per-rank shares."""
import sys
sys.path.insert(0, '.')
import apex_solver_amd as pkg
from apex_solver_amd.solver import GpuSchurComplementSolver, OptimizationType, Problem
name = sys.argv[1] if len(sys.argv) > 1 else 'final-13682'
d = pkg.synthetic.make_named(name)
prob = Problem.bundle_adjustment(d, OptimizationType.SelfCalibration, 1.0)
for world in (2, 4, 8):
    fr = []
    for r in range(world):
        s = GpuSchurComplementSolver(0).with_shard(r, world)
        s.initialize_structure(prob)
        i = s.info()
        fr.append(round(i["dist_local_fraction"], 4))
        top = i["dist_top_columns"]; lev = i["etree_levels"]; nt = i["tile_rows"]
        ops = (i["n_potrf"], i["n_trsm"], i["n_update"])
        s.close()
        if r == 0: ops0 = ops
    print(name, "world", world, "tile rows", nt, "top columns", top, "groups", lev, "fractions", fr, "rank0 ops", ops0, flush=True)
