"""The multi-GPU schedule with N > 1 ranks ON ONE GPU: N processes, the host shared-memory transport (csrc/comm.h,
apexgpu_comm_init_shm) in place of RCCL.  Everything the library does differently for world > 1 -- landmark sharding along
the elimination tree, reduce-to-owner / all-reduce of the partial S, the distributed Cholesky with its summed top tiles and
max-reduced pivot flag, the phased sweeps with their two vector exchanges and the max-reduced sweep time-out word, sharded
back-substitution / cost / statistics, the gather of the owners' points -- runs with the real kernels and is compared with
the single-rank solve of the same system.  (tests/test_gpu_rccl_ranks.py is the same with RCCL where the node has the GPUs;
the RCCL transport itself is eleven thin calls, csrc/comm.cpp.)"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import apex_solver_amd as pkg
from apex_solver_amd.solver import GpuSchurComplementSolver, OptimizationType, Problem, SchurVariant

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    return float(np.linalg.norm(np.ravel(a) - np.ravel(b)) / np.linalg.norm(np.ravel(b)))


def run_ranks(world, variant, tmp_path):
    name = f"{os.getpid()}-{world}-{variant}"
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shm_worker.py"), str(r), str(world), name, str(tmp_path), variant],
                              cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=900) for p in procs]
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r}: {se[-3000:]}"
    return [json.load(open(tmp_path / f"out_{r}.json")) for r in range(world)]


def single_rank(variant):
    d = pkg.synthetic.make_problem(1500, 30000, 3, 7, config_id=310)
    prob = Problem.bundle_adjustment(d, OptimizationType.SelfCalibration, 1.0)
    s = GpuSchurComplementSolver(0)
    if variant == "implicit":
        s.with_variant(SchurVariant.Implicit).with_cg_params(500, 1e-9)
    s.initialize_structure(prob)
    s.set_parameters(d.poses, d.intr, d.points)
    c0 = s.compute_cost()
    step = s.solve_augmented_equation(1e-3)
    gn, sn, pred = s.step_stats()
    c1 = s.eval_step(); s.commit_step()
    q = s.get_parameters()
    pcg = s.info()["pcg_iterations"]
    s.close()
    return dict(c0=c0, step=step, gn=gn, sn=sn, pred=pred, c1=c1, poses=q[0], pts=q[2], nc=prob.layout.cam_dof, n_pt=d.n_pt, pcg=pcg)


@pytest.mark.parametrize("world", [2, 3, 4])
def test_shm_ranks_match_single_rank(world, tmp_path):
    res = run_ranks(world, "sparse", tmp_path)
    ref = single_rank("sparse")
    nc = ref["nc"]
    for mode in ("tree", "range", "replicated"):
        cams = [np.load(tmp_path / f"cam_{mode}_{r}.npy") for r in range(world)]
        for r in range(1, world):   # the replicated / assembled camera step is bit-identical on every rank
            assert np.array_equal(cams[r], cams[0]), (mode, r)
        z = np.load(tmp_path / f"res_{mode}_0.npz")
        o = res[0][mode]
        errs = dict(cost=abs(o["c0"] - ref["c0"]) / ref["c0"], cam_step=rel(z["step"][:nc], ref["step"][:nc]), grad=abs(o["gn"] - ref["gn"]) / ref["gn"],
                    step_norm=abs(o["sn"] - ref["sn"]) / ref["sn"], pred=abs(o["pred"] - ref["pred"]) / abs(ref["pred"]),
                    trial=abs(o["c1"] - ref["c1"]) / ref["c1"], poses=rel(z["poses"], ref["poses"]), points=rel(z["pts"], ref["pts"]))
        print(world, mode, {k: f"{v:.1e}" for k, v in errs.items()}, {k: o["info"][k] for k in ("dist_top_columns", "tree_sharded", "dist_local_fraction")})
        assert errs["cost"] < 1e-13 and errs["grad"] < 1e-11 and errs["cam_step"] < 1e-8 and errs["step_norm"] < 1e-8
        assert errs["pred"] < 1e-6 and errs["trial"] < 1e-8 and errs["poses"] < 1e-9 and errs["points"] < 1e-8
        # every landmark is owned by exactly one rank; no sweep timed out; every rank reports the same scalars
        assert sum(res[r][mode]["owned"] for r in range(world)) == ref["n_pt"]
        for r in range(world):
            assert res[r][mode]["counters"]["sweep_timeouts"] == 0
            for k in ("c0", "gn", "sn", "pred", "c1"):
                assert res[r][mode][k] == res[0][mode][k], (mode, r, k)
    assert res[0]["tree"]["info"]["tree_sharded"] and res[0]["tree"]["info"]["dist_top_columns"] > 0
    assert not res[0]["range"]["info"]["tree_sharded"] and res[0]["replicated"]["info"]["dist_top_columns"] == 0


def test_shm_ranks_matrix_free_variant(tmp_path):
    """IterativeSchurSolver semantics sharded over 2 ranks: g_red, g_c and the Schur-Jacobi blocks all-reduced once, every
    S p once per PCG iteration."""
    world = 2
    res = run_ranks(world, "implicit", tmp_path)
    ref = single_rank("implicit")
    nc = ref["nc"]
    z = np.load(tmp_path / "res_implicit_0.npz")
    o = res[0]["implicit"]
    print("implicit", rel(z["step"][:nc], ref["step"][:nc]), o["info"]["pcg_iterations"], ref["pcg"])
    # (two PCG runs that sum their partial S p in different orders stop within an iteration or two of each other and
    # differ by the tolerance's worth of step)
    assert abs(o["c0"] - ref["c0"]) / ref["c0"] < 1e-13 and rel(z["step"][:nc], ref["step"][:nc]) < 1e-5
    assert abs(o["info"]["pcg_iterations"] - ref["pcg"]) <= max(3, ref["pcg"] // 20)
    assert abs(o["c1"] - ref["c1"]) / ref["c1"] < 1e-6 and rel(z["pts"], ref["pts"]) < 1e-6
    assert np.array_equal(np.load(tmp_path / "cam_implicit_1.npy"), np.load(tmp_path / "cam_implicit_0.npy"))
