"""Worker of tests/test_gpu_rccl_ranks.py: one rank of an N-rank RCCL run (launched by torch.distributed.run).
Every rank solves its shard of ONE damped system through the production path (apexgpu_comm_init + apexgpu_solve_augmented:
ncclAllReduce / ncclReduce inside the library); rank 0 also solves the whole system alone and compares."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def main():
    import torch
    import torch.distributed as dist

    import apex_solver_amd as pkg
    from apex_solver_amd.solver import GpuSchurComplementSolver, OptimizationType, Problem

    rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"]); local = int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    d = pkg.synthetic.make_problem(1500, 30000, 3, 7, config_id=310)
    prob = Problem.bundle_adjustment(d, OptimizationType.SelfCalibration, 1.0)
    lam = 1e-3
    out = {}
    for name, opts in (("tree", {}), ("range", {"tree_sharding": 0}), ("replicated", {"dist_factor": 0})):
        uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            buf = (C.c_char * 128)()
            assert pkg.capi.load().apexgpu_get_unique_id(C.cast(buf, C.c_void_p)) == 0
            uid = torch.frombuffer(bytearray(bytes(buf)), dtype=torch.uint8).cuda()
        dist.broadcast(uid, 0)
        s = GpuSchurComplementSolver(local).with_communicator(world, rank, bytes(uid.cpu().numpy().tobytes()))
        for k, v in opts.items():
            s.with_option(k, v)
        s.initialize_structure(prob)
        s.set_parameters(d.poses, d.intr, d.points)
        c0 = s.compute_cost()
        step = s.solve_augmented_equation(lam)
        gn, sn, pred = s.step_stats()
        c1 = s.eval_step()
        s.commit_step()
        poses, intr, pts = s.get_parameters()
        cam = torch.from_numpy(step[: prob.layout.cam_dof].copy()).cuda()
        cam0 = cam.clone(); dist.broadcast(cam0, 0)
        same = bool(torch.equal(cam, cam0))
        flags = torch.tensor([int(same)], device="cuda"); dist.all_reduce(flags, op=dist.ReduceOp.MIN)
        if rank == 0:
            s1 = GpuSchurComplementSolver(local).initialize_structure(prob)
            s1.set_parameters(d.poses, d.intr, d.points)
            r0 = s1.compute_cost()
            step1 = s1.solve_augmented_equation(lam)
            g1, n1, p1 = s1.step_stats()
            t1 = s1.eval_step(); s1.commit_step()
            q = s1.get_parameters()
            rel = lambda a, b: float(np.linalg.norm(np.ravel(a) - np.ravel(b)) / np.linalg.norm(np.ravel(b)))
            nc = prob.layout.cam_dof
            out[name] = dict(cost=abs(c0 - r0) / r0, cam_step=rel(step[:nc], step1[:nc]), grad_norm=abs(gn - g1) / g1,
                             step_norm=abs(sn - n1) / n1, pred=abs(pred - p1) / abs(p1), trial=abs(c1 - t1) / t1,
                             poses=rel(poses, q[0]), points=rel(pts, q[2]), identical_camera_step_on_all_ranks=bool(flags.item()),
                             info=s.info())
            s1.close()
        s.close()
    if rank == 0:
        print("RCCL_RESULT " + json.dumps(out))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
