"""Worker of tests/test_gpu_shm_ranks.py: ONE rank of an N-rank run of the production multi-GPU schedule over the host
shared-memory transport (csrc/comm.h, apexgpu_comm_init_shm).  The ranks are plain processes that share GPU 0, so the
world > 1 branches of the library -- sharded assembly, reduce-to-owner / all-reduce of S, distributed factorisation with its
summed top tiles and collective pivot flag, phased triangular sweeps, sharded back-substitution, cost and step statistics,
the gather of the owners' points -- all execute with the real kernels on a single-GPU box.
  python tests/shm_worker.py <rank> <world> <run name> <out dir> <variant: sparse|implicit>"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def main():
    rank, world, name, out_dir, variant = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    import apex_solver_amd as pkg
    from apex_solver_amd.solver import GpuSchurComplementSolver, OptimizationType, Problem, SchurVariant

    d = pkg.synthetic.make_problem(1500, 30000, 3, 7, config_id=310)
    prob = Problem.bundle_adjustment(d, OptimizationType.SelfCalibration, 1.0)
    lam = 1e-3
    out = {}
    modes = (("tree", {}), ("range", {"tree_sharding": 0}), ("replicated", {"dist_factor": 0})) if variant == "sparse" else (("implicit", {}),)
    for k, (mode, opts) in enumerate(modes):
        s = GpuSchurComplementSolver(0).with_shm_communicator(world, rank, f"{name}-{k}")
        if variant == "implicit":
            s.with_variant(SchurVariant.Implicit).with_cg_params(500, 1e-9)
        for o, val in opts.items():
            s.with_option(o, val)
        s.initialize_structure(prob)
        s.set_parameters(d.poses, d.intr, d.points)
        c0 = s.compute_cost()
        step = s.solve_augmented_equation(lam)
        gn, sn, pred = s.step_stats()
        c1 = s.eval_step()
        s.commit_step()
        poses, intr, pts = s.get_parameters()
        nc = prob.layout.cam_dof
        np.save(os.path.join(out_dir, f"cam_{mode}_{rank}.npy"), step[:nc])
        out[mode] = dict(c0=c0, gn=gn, sn=sn, pred=pred, c1=c1, info=s.info(), counters=s.counters(),
                         owned=int(s.owned_landmarks().sum()))
        np.savez(os.path.join(out_dir, f"res_{mode}_{rank}.npz"), step=step, poses=poses, intr=intr, pts=pts)
        s.close()
    json.dump(out, open(os.path.join(out_dir, f"out_{rank}.json"), "w"))


if __name__ == "__main__":
    main()
