"""Host-side checks (no GPU) of the elimination-tree cut behind the distributed Cholesky and the tree sharding of the
landmarks (csrc/tile_plan.hip: partition_columns)."""
import numpy as np
import pytest

from apex_solver_amd import capi


def banded(nt, bw):
    p = np.zeros((nt, nt), dtype=np.uint8)
    for i in range(nt):
        p[i, max(0, i - bw):i + 1] = 1
    return p


def nd_order(lo, hi, leaf, sep, out):
    """nested dissection of a path lo..hi-1 with separators of `sep` nodes (what TilePlan::order does to a banded S)"""
    n = hi - lo
    if n <= leaf:
        out.extend(range(lo, hi)); return
    mid = lo + (n - sep) // 2
    nd_order(lo, mid, leaf, sep, out)
    nd_order(mid + sep, hi, leaf, sep, out)
    out.extend(range(mid, mid + sep))


def structure(nt, bw, leaf):
    order = []
    nd_order(0, nt, leaf, bw, order)
    perm = np.empty(nt, dtype=int); perm[order] = np.arange(nt)      # old -> new
    b = banded(nt, bw)
    p = np.zeros_like(b)
    for i in range(nt):
        for j in range(max(0, i - bw), i + 1):
            a, c = perm[i], perm[j]
            p[max(a, c), min(a, c)] = 1
    return p, perm


def symbolic(p):
    nt = p.shape[0]
    rows = [set(np.flatnonzero(p[k + 1:, k]) + k + 1) for k in range(nt)]
    parent = [-1] * nt
    for k in range(nt):
        if rows[k]:
            parent[k] = min(rows[k])
            rows[parent[k]] |= rows[k] - {parent[k]}
    return rows, parent


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_cut_invariants(world):
    p, _ = structure(200, 4, 16)
    owner, n_top = capi.tile_partition(p, world)
    assert n_top > 0 and n_top == int((owner < 0).sum())
    rows, parent = symbolic(p)
    nt = len(owner)
    for k in range(nt):
        if owner[k] < 0:                       # the top is closed under "parent"
            assert parent[k] < 0 or owner[parent[k]] < 0
        else:                                  # a column's rows are its ancestors: same owner or top
            assert all(owner[i] == owner[k] or owner[i] < 0 for i in rows[k])
    assert set(owner[owner >= 0]) == set(range(world))           # every rank owns something
    # work balance below the top (potrf + panel + update counts, the partitioner's own measure)
    w = np.array([1 + len(r) + len(r) * (len(r) + 1) / 2 for r in rows])
    load = np.array([w[owner == r].sum() for r in range(world)])
    assert load.max() / load.sum() < (0.62 if world == 3 else 1.45 / world)
    # deterministic: every rank computes the same cut
    owner2, _ = capi.tile_partition(p, world)
    assert np.array_equal(owner, owner2)


def test_clique_lies_on_one_root_path():
    """Tree sharding relies on it: the tile columns of any clique of S (the cameras of one landmark) below the top belong
    to ONE rank."""
    p, perm = structure(200, 4, 16)
    owner, _ = capi.tile_partition(p, 4)
    for start in range(0, 196):                      # a landmark seen inside a window of the band = a clique
        tiles = perm[start:start + 5]
        owners = {int(owner[t]) for t in tiles if owner[t] >= 0}
        assert len(owners) <= 1


def test_no_cut_for_one_rank_or_a_chain():
    p, _ = structure(64, 4, 16)
    owner, n_top = capi.tile_partition(p, 1)
    assert n_top == 0 and np.all(owner == 0)
    owner, n_top = capi.tile_partition(banded(40, 4), 4)   # natural band order: the tree is one chain, nothing to share out
    assert n_top == 0
