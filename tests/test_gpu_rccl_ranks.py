"""The production multi-GPU path -- one process per GPU, apexgpu_comm_init, RCCL collectives inside the library -- with
2 (and, where the node has them, 4 / 8) ranks against the single-rank solve of the same system, in all three modes
(tree-sharded distributed Cholesky, range-sharded, replicated factorisation = "all-reduce(S)" as north_star words it).
Skips on a 1-GPU box; the single-process lockstep tests (tests/test_gpu_dist_factor.py, tests/test_gpu_configs.py) cover
the same phases there."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gpus():
    import torch

    return torch.cuda.device_count()   # does not initialise the GPU runtime


@pytest.mark.parametrize("world", [2, 4, 8])
def test_rccl_ranks_match_single_rank(world):
    if _gpus() < world:
        pytest.skip(f"needs {world} GPUs on this node")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # the first multi-GPU record of this path should be diagnosable from its log alone: RCCL reports the topology it found,
    # the rings / trees / channels it built and the transport of every connection (xGMI = P2P/direct pointer) at init
    env.setdefault("NCCL_DEBUG", "INFO"); env.setdefault("NCCL_DEBUG_SUBSYS", "INIT,GRAPH,ENV")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "rccl_worker.py")],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    topo = [l for l in (p.stdout + p.stderr).splitlines()
            if "NCCL INFO" in l and any(k in l for k in ("Channel", "Ring", "Tree", "comm 0x", "via P2P", "via SHM", "via NET", "XGMI", "nRanks", "Connected all"))]
    print(f"RCCL topology lines ({len(topo)}):")
    for l in topo[:80]:
        print("   ", l[-200:])
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RCCL_RESULT ")]
    assert line, p.stdout[-2000:]
    res = json.loads(line[0][len("RCCL_RESULT "):])
    for mode, r in res.items():
        print(world, mode, {k: v for k, v in r.items() if k != "info"})
        assert r["identical_camera_step_on_all_ranks"]
        assert r["cost"] < 1e-13 and r["grad_norm"] < 1e-11 and r["cam_step"] < 1e-8 and r["step_norm"] < 1e-8
        assert r["pred"] < 1e-6 and r["trial"] < 1e-8 and r["poses"] < 1e-9 and r["points"] < 1e-8
    assert res["tree"]["info"]["tree_sharded"] and res["tree"]["info"]["dist_top_columns"] > 0
    assert not res["range"]["info"]["tree_sharded"] and res["replicated"]["info"]["dist_top_columns"] == 0
