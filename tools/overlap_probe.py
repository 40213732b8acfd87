#!/usr/bin/env python3
"""What would staggering buy?  Two independent half-size problems (final-13682 at scale 0.5) on two handles / two streams:
one after the other, and from two host threads at once (random phase: assembly of one beside the factorisation of the
other).  The ratio bounds what a staggered schedule of two subtrees could gain on one problem.
  python tools/overlap_probe.py [scale=0.5] [iterations=30]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import apex_solver_amd as pkg
from apex_solver_amd.solver import GpuSchurComplementSolver, OptimizationType, Problem

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 30
d = pkg.datasets.load_named("final-13682", scale)[0]
prob = Problem.bundle_adjustment(d, OptimizationType.SelfCalibration, 1.0)
hs = []
for _ in range(2):
    s = GpuSchurComplementSolver(0)
    for kv in os.environ.get("APEX_PROBE_OPTS", "").split(","):
        if "=" in kv:
            s.with_option(kv.split("=")[0], int(kv.split("=")[1]))
    s.initialize_structure(prob)
    s.set_parameters(d.poses, d.intr, d.points)
    for _ in range(3):
        s.solve_augmented_equation(1e-3, want_step=False)
    hs.append(s)


def loop(s, n):
    for _ in range(n):
        s.solve_augmented_equation(1e-3, want_step=False)


for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loop(hs[0], n_it); loop(hs[1], n_it)
    torch.cuda.synchronize(); t_seq = (time.perf_counter() - t0) * 1e3 / n_it
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=loop, args=(h, n_it)) for h in hs]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize(); t_par = (time.perf_counter() - t0) * 1e3 / n_it
    print(os.environ.get("APEX_PROBE_OPTS", ""), "GPU_MAX_HW_QUEUES=" + os.environ.get("GPU_MAX_HW_QUEUES", "-"), f"scale {scale}: one after the other {t_seq:.2f} ms per pair of solves, two threads {t_par:.2f} ms ({t_seq / t_par:.2f}x)", flush=True)
hs[0].enable_stage_timing(True); hs[0].reset_stage_times(); loop(hs[0], 5)
print({k: round(v[0] / 5, 3) for k, v in hs[0].stage_times().items()})
