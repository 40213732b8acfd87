"""The C ABI driven from plain C (tests/capi_client.c, compiled with gcc against include/apexgpu.h and linked to
libapexgpu.so): the boundary a compiled host binds, without Python in the loop.  Results must equal the Python
binding's (same library, same inputs)."""
import json
import os
import subprocess

import numpy as np
import pytest

import apex_solver_amd as pkg
from apex_solver_amd.pose_graph import PoseGraphProblem
from apex_solver_amd.solver import (LevenbergMarquardt, LevenbergMarquardtConfig, LinearSolverType, OptimizationType, Problem)

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def client():
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "capi_client")
    libdir = os.path.join(ROOT, "apex-solver_amd")
    from oracle import oracle as ora

    ora.lib()   # builds oracle/libba_oracle.so if needed: the client's host-side retraction (level-1 sequence) is the CPU path
    oradir = os.path.join(ROOT, "oracle")
    subprocess.run(["gcc", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "capi_client.c"),
                    "-o", exe, "-L", libdir, "-lapexgpu", "-L", oradir, "-lba_oracle", "-lm", f"-Wl,-rpath,{libdir}",
                    f"-Wl,-rpath,{oradir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def run(exe, kind, path):
    p = subprocess.run([exe, kind, str(path)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    return json.loads(p.stdout.strip().splitlines()[-1])


def test_ba_from_c(client, tmp_path):
    d = pkg.synthetic.make_problem(40, 2500, 3, 7, config_id=91)
    f = tmp_path / "ba.bin"
    with open(f, "wb") as fh:
        np.array([d.n_cam, d.n_pt, d.n_obs, 1], np.int64).tofile(fh)
        d.cam_idx.astype(np.uint32).tofile(fh); d.pt_idx.astype(np.uint32).tofile(fh)
        d.obs_uv.astype(np.float64).tofile(fh); d.poses.tofile(fh); d.intr.tofile(fh); d.points.tofile(fh)
    got = run(client, "ba", f)
    prob = Problem.bundle_adjustment(d, OptimizationType.SelfCalibration, 1.0)
    ref = LevenbergMarquardt.with_config(LevenbergMarquardtConfig.new().with_max_iterations(20)).optimize(prob)
    assert got["status"] == ref.status.value and got["iterations"] == ref.iterations
    assert abs(got["initial_cost"] - ref.initial_cost) <= 1e-12 * ref.initial_cost
    assert abs(got["final_cost"] - ref.final_cost) <= 1e-6 * ref.final_cost


@pytest.mark.parametrize("mode", [1, 0], ids=["selfcal", "ba"])
def test_level1_binding_sequence_reproduces_the_device_loop(client, tmp_path, mode):
    """The call sequence of the Rust binding (rust/src/linearizer/gpu/mod.rs, rust/src/linalg/gpu_schur.rs) under the
    reference's unchanged loop -- apexgpu_set_params in EVERY assemble, apexgpu_solve_augmented, host-side step
    statistics and retraction, a residual evaluation at the trial point -- against the device-resident loop
    (apexgpu_lm_optimize): same status, same iteration count, same costs."""
    d = pkg.synthetic.make_problem(40, 2500, 3, 7, config_id=91)
    f = tmp_path / "ba.bin"
    with open(f, "wb") as fh:
        np.array([d.n_cam, d.n_pt, d.n_obs, mode], np.int64).tofile(fh)
        d.cam_idx.astype(np.uint32).tofile(fh); d.pt_idx.astype(np.uint32).tofile(fh)
        d.obs_uv.astype(np.float64).tofile(fh); d.poses.tofile(fh); d.intr.tofile(fh); d.points.tofile(fh)
    a = run(client, "ba", f)
    b = run(client, "ba-level1", f)
    print("device loop", a, "| level-1 sequence", b)
    assert a["status"] == b["status"] and a["iterations"] == b["iterations"]
    assert abs(a["initial_cost"] - b["initial_cost"]) <= 1e-12 * a["initial_cost"]
    assert abs(a["final_cost"] - b["final_cost"]) <= 1e-7 * a["final_cost"]


def test_pose_graph_from_c(client, tmp_path):
    d = pkg.synthetic.make_sphere(10, 14, id_stride=5)
    f = tmp_path / "pg.bin"
    with open(f, "wb") as fh:
        np.array([d.n_v, d.n_e], np.int64).tofile(fh)
        d.ids.astype(np.int64).tofile(fh); d.e_from.astype(np.uint32).tofile(fh); d.e_to.astype(np.uint32).tofile(fh)
        d.meas.astype(np.float64).tofile(fh); d.poses.astype(np.float64).tofile(fh)
    got = run(client, "pg", f)
    cfg = LevenbergMarquardtConfig.new().with_max_iterations(30).with_linear_solver_type(LinearSolverType.SparseCholesky)
    ref = LevenbergMarquardt.with_config(cfg).optimize(PoseGraphProblem.pose_graph(d))
    assert got["status"] == ref.status.value and got["iterations"] == ref.iterations
    assert abs(got["final_cost"] - ref.final_cost) <= 1e-8 * ref.final_cost
