"""bench.py's one-line JSON contract (metric / value / roofline / cpu_baseline ...) on a small workload."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def run(*args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_ba_line():
    out = run("--workload", "ladybug-49", "--steps", "3", "--warmup", "1")
    for k in REQUIRED:
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["warmup"] == 1 and out["higher_is_better"] is False
    assert out["dtype"] == "f64" and out["data"] == "synthetic" and out["vs_baseline"] is None and "workload" in out["config"]
    assert out["value"] == pytest.approx(out["ms_per_step"]) and out["value"] > 0
    r = out["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["peak"] == 8000.0 and r["frac"] == pytest.approx(r["achieved"] / r["peak"])
    c = out["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert out["final_cost"] < out["initial_cost"]
    # round 5: the factorisation against ITS roof, the variant that really ran, and the fall-back figure that does not depend
    # on the state of the handle (one PCG iteration x the reference's cap) next to the observed one
    f = out["factor"]
    assert f["bound"] == "mfma" and f["peak"] == 78.6 and f["frac"] == pytest.approx(f["achieved"] / 78.6) and f["flops_per_factorisation"] > 0
    assert out["config"]["variant_used"] == "Sparse" and out["config"]["reason"] == ""
    fb = out["fallback_implicit"]
    assert fb["cap"] == 500 and fb["ms_per_pcg_iter"] > 0 and fb["bound_ms"] >= fb["ms_per_pcg_iter"] * 500 and len(fb["observed_iterations"]) == 2
    assert "other_workloads" not in out          # only the headline line carries them


def test_pose_graph_line():
    out = run("--workload", "sphere2500", "--scale", "0.04", "--steps", "3", "--warmup", "1")
    for k in REQUIRED:
        assert k in out, k
    assert out["roofline"]["bound"] == "mfma" and out["scaling"] == "weak" and out["final_cost"] < out["initial_cost"]


def test_two_ranks_fall_back_to_shared_memory_when_rccl_refuses():
    """`--gpus 2` with both ranks on the one GPU of the box (gloo bring-up mode): RCCL refuses the duplicate device, the ranks
    vote, and the run completes over the shared-memory transport -- the line says which transport carried it."""
    env = dict(os.environ, APEX_BENCH_PG="gloo")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "ladybug-1723", "--scale", "0.25", "--steps", "3",
                        "--warmup", "1", "--no-cpu-baseline", "--no-other-variants"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["final_cost"] < out["initial_cost"]
    assert out["config"]["transport"].startswith("shm (RCCL communicator failed"), out["config"]["transport"]
