"""Host-only checks of the sorted camera-pair lists the default Schur reduction consumes (csrc/schur_pairs.h): the
kernel's bookkeeping -- chunk masks, block-local indices, padding, flush points, task cuts -- is replayed in numpy and
must deliver every camera pair (i, j) of every landmark to the block S(cam_i, cam_j) exactly once.  No GPU."""
import numpy as np
import pytest

import apex_solver_amd as pkg
from apex_solver_amd import capi

PAD = 0xFFFFFFFF
NB = 144


def replay(pl, cam_idx, dc):
    """What k_schur_pairs does with the lists, minus the arithmetic: {block index: [(i, j), ...]} as flushed."""
    recs, chunks, blocks, tasks, o_index = pl["recs"], pl["chunks"], pl["blocks"], pl["tasks"], pl["o_index"]
    flushed = {}
    seen_chunks = np.zeros(len(chunks), dtype=int)
    for c0, n in tasks:
        cur = -1
        acc = []
        for ch in range(c0, c0 + n):
            seen_chunks[ch] += 1
            mask, first = int(np.uint32(chunks[ch, 0])), int(chunks[ch, 1])
            nblk = 1 + bin(mask & ~1).count("1")
            assert nblk <= 32
            for s in range(32):
                if (mask >> s) & 1:
                    if cur >= 0:
                        flushed.setdefault(cur, []).extend(acc)
                    cur = first if cur < 0 else cur + 1
                    acc = []
                for p in (64 * ch + 2 * s, 64 * ch + 2 * s + 1):
                    i, j, l, blk = (int(x) for x in recs[p])
                    if i == PAD:
                        continue
                    assert cur >= 0 and first + blk == cur, "a lane's block-local index must name the block being accumulated"
                    assert blk < nblk
                    acc.append((i, j, l))
            if ch == c0:
                assert mask & 1, "a task starts at a block boundary"
        if cur >= 0:
            flushed.setdefault(cur, []).extend(acc)
    assert (seen_chunks == 1).all(), "every chunk belongs to exactly one task"
    return flushed


@pytest.mark.parametrize("dc", [9, 6])
@pytest.mark.parametrize("shuffled", [False, True], ids=["grouped", "shuffled"])
@pytest.mark.parametrize("shape", [(40, 1500, 3, 7), (300, 9000, 2, 9)])
def test_every_pair_reaches_its_block_once(dc, shape, shuffled):
    n_cam, n_pt, klo, khi = shape
    d = pkg.synthetic.make_problem(n_cam, n_pt, klo, khi, config_id=400 + n_cam)
    if shuffled:   # the generator lists observations landmark by landmark (the set-up's fast path); any order must do
        perm = np.random.default_rng(7).permutation(d.n_obs)
        d.cam_idx, d.pt_idx, d.obs_uv = d.cam_idx[perm], d.pt_idx[perm], d.obs_uv[perm]
    pl = capi.pair_lists(d.n_cam, d.n_pt, dc, d.cam_idx, d.pt_idx)
    flushed = replay(pl, d.cam_idx, dc)
    blocks, o_index = pl["blocks"], pl["o_index"]
    cpt = NB // dc
    got = {}
    for b, pairs in flushed.items():
        dst, ci, cj, flags = (int(x) for x in blocks[b])
        I, J = ci // cpt, cj // cpt
        assert dst == (I * (I + 1) // 2 + J) * NB * NB + (ci % cpt) * dc * NB + (cj % cpt) * dc
        assert cj <= ci and ((flags & 2) != 0) == (ci == cj)
        for i, j, l in pairs:
            oi, oj = o_index[i], o_index[j]
            assert d.cam_idx[oi] == ci and d.cam_idx[oj] == cj and d.pt_idx[oi] == l and d.pt_idx[oj] == l
            key = (min(oi, oj), max(oi, oj))
            assert key not in got
            got[key] = b
    # expected: all unordered pairs of observations of one landmark
    order = np.lexsort((d.cam_idx, d.pt_idx))
    ptr = np.concatenate([[0], np.cumsum(np.bincount(d.pt_idx, minlength=d.n_pt))])
    want = 0
    for l in range(d.n_pt):
        k = ptr[l + 1] - ptr[l]
        want += k * (k - 1) // 2
    assert len(got) == want
    # blocks are visited once unless flagged for an atomic flush, rows in the caller's camera order
    keys = [(int(blocks[b][1]), int(blocks[b][2])) for b in sorted(flushed)]
    assert keys == sorted(keys)
    plain = [k for b, k in zip(sorted(flushed), keys) if int(blocks[b][3]) == 0]
    assert len(plain) == len(set(plain))


def test_hub_blocks_are_split_and_flagged_atomic():
    """Two cameras that share 20,000 landmarks: their block exceeds what one wave takes and is cut into pieces that
    flush with atomic adds; a camera that sees a landmark twice lands on the diagonal block with the B + B^T flag."""
    n_cam, n_pt = 12, 20000
    rng = np.random.default_rng(0)
    cam_idx, pt_idx = [], []
    for l in range(n_pt):
        cams = [3, 7] + ([int(rng.integers(8, 12))] if l % 5 == 0 else [])
        if l == 17:
            cams = [5, 5, 9]
        cam_idx += cams; pt_idx += [l] * len(cams)
    cam_idx = np.asarray(cam_idx, dtype=np.uint32); pt_idx = np.asarray(pt_idx, dtype=np.uint32)
    pl = capi.pair_lists(n_cam, n_pt, 9, cam_idx, pt_idx)
    flushed = replay(pl, cam_idx, 9)
    blocks = pl["blocks"]
    hub = [b for b in flushed if (int(blocks[b][1]), int(blocks[b][2])) == (7, 3)]
    assert len(hub) >= 3 and all(int(blocks[b][3]) & 1 for b in hub)
    assert sum(len(flushed[b]) for b in hub) == n_pt - 1
    assert len({int(blocks[b][0]) for b in hub}) == 1                  # all pieces add into the same block
    diag = [b for b in flushed if int(blocks[b][1]) == int(blocks[b][2])]
    assert len(diag) == 1 and int(blocks[diag[0]][3]) & 2 and len(flushed[diag[0]]) == 1


def test_empty_and_single_observation_landmarks():
    cam_idx = np.array([0, 1, 2, 2], dtype=np.uint32); pt_idx = np.array([0, 0, 3, 5], dtype=np.uint32)
    pl = capi.pair_lists(4, 7, 9, cam_idx, pt_idx)
    flushed = replay(pl, cam_idx, 9)
    assert sum(len(v) for v in flushed.values()) == 1 and len(pl["tasks"]) == 1
    pl0 = capi.pair_lists(4, 7, 6, np.array([2], dtype=np.uint32), np.array([1], dtype=np.uint32))
    assert len(pl0["tasks"]) == 0 and len(pl0["recs"]) == 0


# ---- the queued layout ("schur_form" 4): every lane group owns a block of its own ------------------------------------------
def replay_queued(pl):
    """What k_schur_pairs_r<.., QL = true> does with the lists, minus the arithmetic: group g of a wave adds the pairs
    g, g + 7, ..., g + 56 of every chunk of its task and, where the chunk's flush bit g is set, hands what it has gathered
    to the destination of ITS descriptor of that chunk.  Returns [(dst, cj, flags, [(i, j, l), ...]), ...] per flush."""
    recs, chunks, tasks, qd = pl["recs"], pl["chunks"], pl["tasks"], pl["qdesc"]
    NQ, QLEN = (7, 9) if pl.get("dc", 9) == 9 else (16, 4)   # queues per chunk, pairs of a queue per chunk (round 5: six-column cameras)
    out = []
    seen = np.zeros(len(chunks), dtype=int)
    for c0, n in tasks:
        assert 1 <= n <= 64, "a task's chunk descriptors are held one per lane"
        acc = [[] for _ in range(NQ)]
        carry = [None] * (NQ + 1)
        ci = int(qd[c0, NQ, 1])
        for ch in range(c0, c0 + n):
            seen[ch] += 1
            assert int(qd[ch, NQ, 1]) == ci, "one row camera per task (1 + NQ cameras staged per chunk)"
            for sl in range(NQ * QLEN, 64):
                assert int(recs[64 * ch + sl, 0]) == PAD
            fm = int(np.uint32(chunks[ch, 0]))
            for g in range(NQ):
                dst, cj, flags = (int(x) for x in qd[ch, g])
                assert ((flags & 4) != 0) == (((fm >> g) & 1) != 0), "the flush bit of the mask and of the descriptor agree"
                for t in range(QLEN):
                    i, j, l, q = (int(x) for x in recs[64 * ch + g + NQ * t])
                    if i != PAD:
                        assert q == g
                        acc[g].append((i, j, l, dst, cj))
                if (fm >> g) & 1:
                    assert all(a[3] == dst and a[4] == cj for a in acc[g]), "a queue holds one block between two flushes"
                    if flags & 8:      # head part of a block cut between queue g - 1 and g: carried to the end of the task
                        assert g > 0 and carry[g] is None and ch < c0 + n - 1
                        carry[g] = (dst, cj, acc[g])
                    else:
                        pairs = [a[:3] for a in acc[g]]
                        if flags & 16:     # tail part: joined with the next queue's carried head, stored once
                            assert ch == c0 + n - 1 and carry[g + 1] is not None and carry[g + 1][:2] == (dst, cj)
                            pairs += [a[:3] for a in carry[g + 1][2]]
                            carry[g + 1] = None
                        out.append((dst, ci, cj, flags & 3, pairs))
                    acc[g] = []
        assert all(not a for a in acc), "every queue ends its task flushed"
        assert all(c is None for c in carry), "every carried head was joined"
    assert (seen == 1).all()
    return out


def check_queued(d_cam_idx, d_pt_idx, n_cam, n_pt, pl):
    o_index = pl["o_index"]
    dc = pl.get("dc", 9)
    cpt = NB // dc
    got = {}
    per_block = {}
    for dst, ci, cj, flags, pairs in replay_queued(pl):
        I, J = ci // cpt, cj // cpt
        assert dst == (I * (I + 1) // 2 + J) * NB * NB + (ci % cpt) * dc * NB + (cj % cpt) * dc
        assert cj <= ci and ((flags & 2) != 0) == (ci == cj)
        per_block.setdefault((ci, cj), []).append(flags)
        for i, j, l in pairs:
            oi, oj = o_index[i], o_index[j]
            assert d_cam_idx[oi] == ci and d_cam_idx[oj] == cj and d_pt_idx[oi] == l and d_pt_idx[oj] == l
            key = (min(oi, oj), max(oi, oj))
            assert key not in got
            got[key] = (ci, cj)
    k = np.bincount(d_pt_idx, minlength=n_pt).astype(np.int64)
    assert len(got) == int((k * (k - 1) // 2).sum())
    # a block flushed more than once (cut between two queues, or longer than a piece) adds atomically every time
    for key, fl in per_block.items():
        if len(fl) > 1:
            assert all(f & 1 for f in fl), key
    return per_block


@pytest.mark.parametrize("dc", [9, 6])
@pytest.mark.parametrize("shuffled", [False, True], ids=["grouped", "shuffled"])
@pytest.mark.parametrize("shape", [(40, 1500, 3, 7), (300, 9000, 2, 9)])
def test_queued_layout_delivers_every_pair_once(shape, shuffled, dc):
    n_cam, n_pt, klo, khi = shape
    d = pkg.synthetic.make_problem(n_cam, n_pt, klo, khi, config_id=400 + n_cam)
    if shuffled:
        perm = np.random.default_rng(7).permutation(d.n_obs)
        d.cam_idx, d.pt_idx, d.obs_uv = d.cam_idx[perm], d.pt_idx[perm], d.obs_uv[perm]
    pl = capi.pair_lists_queued(d.n_cam, d.n_pt, d.cam_idx, d.pt_idx, dc)
    per_block = check_queued(d.cam_idx, d.pt_idx, d.n_cam, d.n_pt, pl)
    # every block of these shapes (none longer than a piece, no camera sees a landmark twice) is stored exactly once with
    # plain stores -- also the blocks cut between two queues of a task (carried head + tail)
    assert all(fl == [0] for fl in per_block.values())
    nq = 7 if dc == 9 else 16
    joins = int(((pl["qdesc"][:, :nq, 2] & 16) != 0).sum())
    assert 0 < joins <= (nq - 1) * len(pl["tasks"])


def test_queued_layout_hub_block_pieces_and_diagonal():
    n_cam, n_pt = 12, 20000
    rng = np.random.default_rng(0)
    cam_idx, pt_idx = [], []
    for l in range(n_pt):
        cams = [3, 7] + ([int(rng.integers(8, 12))] if l % 5 == 0 else [])
        if l == 17:
            cams = [5, 5, 9]
        cam_idx += cams; pt_idx += [l] * len(cams)
    cam_idx = np.asarray(cam_idx, dtype=np.uint32); pt_idx = np.asarray(pt_idx, dtype=np.uint32)
    for dc, piece in ((9, 576), (6, 256)):
        pl = capi.pair_lists_queued(n_cam, n_pt, cam_idx, pt_idx, dc)
        per_block = check_queued(cam_idx, pt_idx, n_cam, n_pt, pl)
        assert len(per_block[(7, 3)]) >= (n_pt - 1) // piece and all(f & 1 for f in per_block[(7, 3)])
        assert per_block[(5, 5)] and all(f & 2 for f in per_block[(5, 5)])


@pytest.mark.parametrize("seed", range(12))
def test_queued_layout_on_random_structures(seed):
    """Random co-visibility: a few cameras that share hundreds to thousands of landmarks (blocks longer than a piece, rows of
    several tasks, queues cut inside blocks), many that share a handful (tasks of a single chunk, empty queues), a camera that
    sees some landmarks twice.  Every pair must arrive once, carried heads must be joined, pieces must add atomically."""
    rng = np.random.default_rng(1000 + seed)
    n_cam = int(rng.integers(4, 40))
    n_pt = int(rng.integers(30, 4000))
    popular = rng.choice(n_cam, size=min(n_cam, int(rng.integers(2, 5))), replace=False)
    cam_idx, pt_idx = [], []
    for l in range(n_pt):
        cams = set(int(c) for c in popular if rng.random() < 0.8)
        for _ in range(int(rng.integers(0, 4))):
            cams.add(int(rng.integers(0, n_cam)))
        cams = sorted(cams)
        if len(cams) < 2:
            cams = sorted({int(popular[0]), int((popular[0] + 1) % n_cam)})
        if rng.random() < 0.01:
            cams.append(cams[0])           # one camera sees the landmark twice: a pair on the diagonal block
        cam_idx += cams; pt_idx += [l] * len(cams)
    cam_idx = np.asarray(cam_idx, dtype=np.uint32); pt_idx = np.asarray(pt_idx, dtype=np.uint32)
    perm = rng.permutation(len(cam_idx))
    cam_idx, pt_idx = cam_idx[perm], pt_idx[perm]
    pl = capi.pair_lists_queued(n_cam, n_pt, cam_idx, pt_idx, 9 if seed % 2 == 0 else 6)
    per_block = check_queued(cam_idx, pt_idx, n_cam, n_pt, pl)
    assert per_block
    # and the first layout delivers the same pairs to the same blocks
    pl3 = capi.pair_lists(n_cam, n_pt, 9, cam_idx, pt_idx)
    flushed = replay(pl3, cam_idx, 9)
    n3 = sum(len(v) for v in flushed.values())
    k = np.bincount(pt_idx, minlength=n_pt).astype(np.int64)
    assert n3 == int((k * (k - 1) // 2).sum())
