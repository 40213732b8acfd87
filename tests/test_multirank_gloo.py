"""The N > 1 path on CPU: two processes (gloo) shard the landmarks exactly as the library does
(apexgpu_shard_range, host arithmetic), form their partial S and g_red, all-reduce them and must
obtain the full reduced camera system -- the exchange step of SURVEY.md §8(e).  The partial sums are
produced by the oracle (the checker); the GPU twin of this test is
tests/test_gpu_parity.py::test_shard_partials_sum_to_full."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, out):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import apex_solver_amd as pkg
    from oracle import oracle as ora

    d = pkg.synthetic.make_problem(24, 900, 3, 7, config_id=330)
    lay = pkg.layout.reference_column_layout(d.n_cam, d.n_pt)
    lam = 1e-3
    lo, hi = pkg.capi.shard_range(d.pt_idx, d.n_pt, rank, world)
    sel = (d.pt_idx >= lo) & (d.pt_idx < hi)
    fix = np.zeros((d.n_cam, 6), dtype=np.uint8); fix[0] = 1
    part = ora.OracleProblem(d.n_cam, d.n_pt, d.cam_idx[sel], d.pt_idx[sel], d.obs_uv[sel], lay.intr_col, lay.pose_col,
                             lay.pt_col, mode="selfcal", huber_delta=1.0, fix_pose=fix)
    part.set_params(d.poses, d.intr, d.points)
    cost_part = part.linearize()[0]
    _, grad, S, gred = part.solve_augmented(lam, 0, want_schur=True)
    if rank != 0:  # lambda*I on the camera block is added once (rank 0), as Solver::assemble does
        S = S - lam * np.eye(S.shape[0])
    tS, tg, tc = torch.from_numpy(S.copy()), torch.from_numpy(gred.copy()), torch.tensor([cost_part], dtype=torch.float64)
    dist.all_reduce(tS); dist.all_reduce(tg); dist.all_reduce(tc)
    counts = torch.tensor([int(sel.sum()), lo, hi]); gathered = [torch.zeros(3, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(gathered, counts)
    if rank == 0:
        full = ora.from_data(d, lay, mode="selfcal")
        cost_full = full.linearize()[0]
        _, _, S_full, g_full = full.solve_augmented(lam, 0, want_schur=True)
        rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
        out.put(dict(S=rel(tS.numpy(), S_full), g=rel(tg.numpy(), g_full), cost=abs(tc.item() - cost_full) / cost_full,
                     ranges=[g.tolist() for g in gathered], n_obs=d.n_obs, n_pt=d.n_pt))
    dist.destroy_process_group()


def test_two_rank_landmark_shards_allreduce_to_full_system():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = out.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (n0, lo0, hi0), (n1, lo1, hi1) = res["ranges"]
    assert lo0 == 0 and hi0 == lo1 and hi1 == res["n_pt"]          # contiguous cover
    assert n0 + n1 == res["n_obs"] and abs(n0 - n1) <= 16          # balanced by observations
    assert res["S"] < 1e-12 and res["g"] < 1e-11 and res["cost"] < 1e-13, res


def _tree_worker(rank, world, port, out):
    """The DEFAULT multi-GPU layout (tree sharding, csrc/ba_structure.h): a landmark belongs to the rank that owns the
    tile columns its cameras touch below the shared top of the elimination tree.  Claim under test: the blocks of S in
    a rank's own tile columns are COMPLETE from that rank's landmarks alone (no reduction), and only the blocks of the
    shared top columns need the all-reduce.  Partial sums from the oracle, ownership from the library's host code."""
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import apex_solver_amd as pkg
    from oracle import oracle as ora

    d = pkg.synthetic.make_problem(640, 9000, 3, 7, config_id=331)       # 40 tile rows: a tree worth cutting
    lay = pkg.layout.reference_column_layout(d.n_cam, d.n_pt)
    lam = 1e-3
    hs = pkg.capi.host_structure(d.n_cam, d.n_pt, d.cam_idx, d.pt_idx, mode=1, rank=rank, world=world)
    assert hs["tree_sharded"] == 1.0 and hs["top_columns"] > 0
    owned, cmap, towner = hs["owned"], hs["cmap"], hs["tile_owner"]
    sel = owned[d.pt_idx]
    fix = np.zeros((d.n_cam, 6), dtype=np.uint8); fix[0] = 1
    part = ora.OracleProblem(d.n_cam, d.n_pt, d.cam_idx[sel], d.pt_idx[sel], d.obs_uv[sel], lay.intr_col, lay.pose_col,
                             lay.pt_col, mode="selfcal", huber_delta=1.0, fix_pose=fix)
    part.set_params(d.poses, d.intr, d.points)
    part.linearize()
    _, _, S, gred = part.solve_augmented(lam, 0, want_schur=True)
    S = S - lam * np.eye(S.shape[0])                                      # lambda is added once, by the column's owner
    # column owner of every reference column of S: camera -> internal camera -> tile column -> owner (-1: shared top)
    cpt = 16
    col_cam = np.empty(9 * d.n_cam, dtype=np.int64)
    for c in range(d.n_cam):
        col_cam[lay.pose_col[c]:lay.pose_col[c] + 6] = c
        col_cam[lay.intr_col[c]:lay.intr_col[c] + 3] = c
    tile_of = cmap[col_cam] // cpt
    # block (a, b) of S lives in tile column min(tile(a), tile(b)) of the lower-triangular tile storage
    colt = np.minimum(tile_of[:, None], tile_of[None, :])
    own = towner[colt]
    mine = own == rank
    top = own < 0
    t_top = torch.from_numpy(np.where(top, S, 0.0)); dist.all_reduce(t_top)          # the only matrix exchange
    assembled = np.where(mine, S, 0.0)
    t_all = torch.from_numpy(assembled.copy()); dist.all_reduce(t_all)               # (test bookkeeping: collect the owners' columns)
    tg = torch.from_numpy(gred.copy()); dist.all_reduce(tg)
    n_owned = torch.tensor([int(owned.sum())]); dist.all_reduce(n_owned)
    if rank == 0:
        full = ora.from_data(d, lay, mode="selfcal")
        full.linearize()
        _, _, S_full, g_full = full.solve_augmented(lam, 0, want_schur=True)
        S_full = S_full - lam * np.eye(S_full.shape[0])
        S_dist = t_all.numpy() + t_top.numpy()
        rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
        out.put(dict(S=rel(S_dist, S_full), g=rel(tg.numpy(), g_full), n_owned=int(n_owned.item()), n_pt=d.n_pt,
                     top_columns=int(hs["top_columns"]), frac_top=float(top.mean())))
    dist.destroy_process_group()


def test_two_rank_tree_sharding_needs_no_reduction_of_owned_columns():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tree_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = out.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    print(res)
    assert res["n_owned"] == res["n_pt"]                       # every landmark on exactly one rank
    assert 0 < res["frac_top"] < 0.5
    assert res["S"] < 1e-12 and res["g"] < 1e-11, res


def test_shard_range_properties():
    import apex_solver_amd as pkg

    d = pkg.synthetic.make_problem(16, 5000, 3, 9, config_id=12)
    for world in (1, 2, 3, 8):
        rs = [pkg.capi.shard_range(d.pt_idx, d.n_pt, r, world) for r in range(world)]
        assert rs[0][0] == 0 and rs[-1][1] == d.n_pt
        assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
        cnt = [int(((d.pt_idx >= lo) & (d.pt_idx < hi)).sum()) for lo, hi in rs]
        assert sum(cnt) == d.n_obs and max(cnt) - min(cnt) <= 2 * 9
    with pytest.raises(pkg.capi.LinAlgError):
        pkg.capi.shard_range(d.pt_idx, d.n_pt, 3, 2)
