#!/usr/bin/env python3
"""A/B timing of the Schur-reduction forms on one workload (stage timers = HIP events on the solver's stream).
  python tools/schur_bench.py [--workload final-13682] [--scale 1.0] [--forms 4,3] [--iters 10] [--mode selfcal]
Prints one line per form: ms per assembly stage, set-up seconds by phase."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("APEX_SYNTH_CACHE", "/tmp/apex_synth_cache")

import numpy as np  # noqa: E402

import apex_solver_amd as pkg  # noqa: E402
from apex_solver_amd.solver import GpuSchurComplementSolver, OptimizationType, Problem  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="final-13682")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--forms", default="4,3")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--mode", default="selfcal")
    ap.add_argument("--check", action="store_true", help="compare S x of every form with the first one's")
    a = ap.parse_args()
    t = time.time()
    d = pkg.synthetic.make_named(a.workload, a.scale)
    print(f"{d.name}: {d.n_cam} cameras / {d.n_pt} landmarks / {d.n_obs} observations (generated in {time.time() - t:.1f} s)", flush=True)
    ot = OptimizationType.SelfCalibration if a.mode == "selfcal" else OptimizationType.BundleAdjustment
    prob = Problem.bundle_adjustment(d, ot, 1.0)
    ref = None
    x = np.random.default_rng(0).normal(size=prob.layout.cam_dof)
    for form in [int(f) for f in a.forms.split(",")]:
        s = GpuSchurComplementSolver(0).with_option("schur_form", form)
        t = time.time()
        s.initialize_structure(prob)
        s.set_parameters(d.poses, d.intr, d.points)
        setup = time.time() - t
        for _ in range(2):
            s.assemble(1e-3)
        s.enable_stage_timing(True); s.reset_stage_times()
        for _ in range(a.iters):
            s.assemble(1e-3)
        st = s.stage_times()
        line = {k: round(v[0] / max(v[1], 1), 3) for k, v in st.items() if v[1] > 0}
        print(f"form {form}: {line}", flush=True)
        print(f"   setup {setup:.2f} s  {s.setup_times()}", flush=True)
        if a.check:
            y, _ = s.schur_matvec(1e-3, x, implicit=False)
            if ref is None:
                ref = y
            else:
                print(f"   S x vs first form: {np.linalg.norm(y - ref) / np.linalg.norm(ref):.2e}", flush=True)
        s.close()


if __name__ == "__main__":
    main()
