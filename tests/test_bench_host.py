"""bench.py's launcher logic on the CPU (no GPU is touched): the first multi-GPU run must not be the first time this
code executes."""
import os
import re
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_more_gpus_than_devices_exits_nonzero_before_touching_a_gpu():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert p.returncode != 0
    assert "--gpus 8" in p.stderr and "GPU(s)" in p.stderr     # spawn_ranks' own message, not a HIP error
    assert "{" not in p.stdout                                   # no JSON line


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr


def test_spawn_ranks_starts_children_and_never_replaces_the_process(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench

    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = list(cmd), dict(env)
        return types.SimpleNamespace(returncode=7)

    import torch
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "1"])
    for name in ("execv", "execve", "execvp", "execvpe", "execl", "execlp"):   # a re-exec would take the GPU box down
        monkeypatch.setattr(os, name, lambda *a, **k: pytest.fail("bench.py must not exec"))
    with pytest.raises(SystemExit) as e:
        bench.spawn_ranks(8)
    assert e.value.code == 7                                       # the children's exit code is passed on
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "8", "--steps", "5", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"


def test_no_exec_anywhere_in_the_product_or_the_bench():
    pat = re.compile(r"\bos\.exec|\bexecv|\bexecl")
    for folder, _, files in os.walk(ROOT):
        if any(part in folder for part in (".git", "gpurun_out", "_build", "__pycache__", "/tests")):
            continue
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(folder, f)).read()
                assert not pat.search(src), os.path.join(folder, f)


def test_usable_cores_respects_affinity_and_quota():
    sys.path.insert(0, ROOT)
    import bench

    n = bench.usable_cores()
    assert 1 <= n <= len(os.sched_getaffinity(0))
