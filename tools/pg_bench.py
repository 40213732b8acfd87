#!/usr/bin/env python3
"""Per-iteration timing of the SE3 pose-graph backend (BASELINE.json configs[1]) with stage breakdown.
  python tools/pg_bench.py [rings per_ring] [--nd LEAF]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import apex_solver_amd as pkg  # noqa: E402
from apex_solver_amd.pose_graph import GpuSparseCholeskySolver, PoseGraphProblem  # noqa: E402


def main():
    args = [a for i, a in enumerate(sys.argv[1:]) if not a.startswith("--") and not sys.argv[i].startswith("--")]
    rings, per = (int(args[0]), int(args[1])) if len(args) >= 2 else (50, 50)
    nd = 2
    if "--nd" in sys.argv:
        nd = int(sys.argv[sys.argv.index("--nd") + 1])
    d = pkg.synthetic.make_sphere(rings, per)
    prob = PoseGraphProblem.pose_graph(d)
    s = GpuSparseCholeskySolver().with_option("nested_dissection", nd)
    for opt in ("update_overlap", "factor_flow", "two_side"):
        if "--" + opt in sys.argv:
            s.with_option(opt, int(sys.argv[sys.argv.index("--" + opt) + 1]))
    t0 = time.perf_counter(); s.initialize_structure(prob); setup = time.perf_counter() - t0
    s.set_parameters(d.poses)
    lam = 1e-3
    cost = s.compute_cost()

    def it():
        nonlocal lam, cost
        s.solve_augmented_equation(lam, want_step=False)
        gn, sn, pred = s.step_stats()
        nc = s.eval_step()
        if cost - nc > 0:
            cost = nc; s.commit_step(); lam = max(lam / 3, 1e-12)
        else:
            s.discard_step(); lam *= 2
    for _ in range(3):
        it()
    s.enable_stage_timing(True); s.reset_stage_times()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        it()
    ms = (time.perf_counter() - t0) * 1e3 / n
    st = s.stage_times()
    print(json.dumps({"workload": d.name, "n_v": d.n_v, "n_e": d.n_e, "ms_per_iter": ms, "setup_s": setup, "info": s.info(),
                      "nd_leaf": nd, "stages_ms": {k: v[0] / n for k, v in st.items()}, "cost": cost}))


if __name__ == "__main__":
    main()
