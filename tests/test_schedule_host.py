"""The factorisation's launch sequence proven race free on the host (no GPU): TilePlan::enqueue_factor records its
launches, event records and stream waits instead of issuing them (a plan built on made-up addresses), and
TilePlan::check_schedule demands a happens-before edge between every two launches that touch one tile with a writer among
them.  Round-3 advice: the U2a / U2b1 / U2b2 split had dropped the edge behind a level WITHOUT side-stream work; the checker
finds that bug on the advisor's own pattern, and the fixed schedule passes it, random structures, banded and dense ones,
every schedule switch, distributed cuts -- with and without the dataflow launch of the top groups."""
import numpy as np
import pytest

import apex_solver_amd as pkg

chk = pkg.capi.check_schedule


def lower(nt, entries):
    p = np.eye(nt, dtype=np.uint8)
    for i, j in entries:
        p[max(i, j), min(i, j)] = 1
    return p


def advisor_pattern():
    """col 0 rows {3, 4}, col 1 {2}, col 2 {3}, col 3 {4}: U2b1 of level 0 updates tile (4, 4) on the side stream, level 1 has
    no U2 work at all, and U1d of level 2 updates (4, 4) on the main stream."""
    return lower(5, [(3, 0), (4, 0), (2, 1), (3, 2), (4, 3)])


def test_checker_finds_the_round3_idle_level_bug_and_the_fix_passes():
    p = advisor_pattern()
    kw = dict(overlap=1, split_u1=1, flood_gate=0, factor_flow=0)   # (update_overlap = 1: every batch goes to the side stream)
    for two_side in (0, 2):
        bad = chk(p, two_side=two_side, old_idle_level_bug=True, **kw)
        good = chk(p, two_side=two_side, **kw)
        assert good["levels"] == 4 and good["violations"] == 0, good
        assert bad["violations"] > 0 and "unordered accesses" in bad["first"], bad


def test_every_stream_wait_of_a_small_schedule_is_either_needed_or_redundant_but_none_is_missing():
    """Dropping one wait at a time: the checker must report violations for some of them (they carry the ordering) and never
    for the intact sequence."""
    rng = np.random.default_rng(3)
    nt = 14
    p = lower(nt, [(i, j) for i in range(nt) for j in range(i) if i - j <= 2 or rng.random() < 0.08])
    base = chk(p, two_side=2, overlap=1, split_u1=1, flood_gate=0, factor_flow=0)
    assert base["violations"] == 0 and base["waits"] >= 6, base
    needed = 0
    for k in range(base["waits"]):
        r = chk(p, two_side=2, overlap=1, split_u1=1, flood_gate=0, factor_flow=0, drop_wait=k)
        assert r["dropped"]
        needed += r["violations"] > 0
    assert needed >= 3, needed


def structures():
    rng = np.random.default_rng(11)
    out = [("advisor", advisor_pattern())]
    for nt, band in ((24, 2), (40, 4), (64, 3)):
        out.append((f"band{nt}", lower(nt, [(i, j) for i in range(nt) for j in range(max(0, i - band), i)])))
    out.append(("dense12", lower(12, [(i, j) for i in range(12) for j in range(i)])))
    out.append(("arrow30", lower(30, [(29, j) for j in range(29)] + [(28, j) for j in range(0, 28, 3)])))
    for seed in range(6):
        nt = int(rng.integers(8, 40))
        dens = float(rng.choice([0.03, 0.08, 0.2]))
        out.append((f"random{seed}", lower(nt, [(i, j) for i in range(nt) for j in range(i) if rng.random() < dens or i - j == 1 and rng.random() < 0.7])))
    # forests and chains that skip levels: a long chain next to short ones
    out.append(("chain+leaves", lower(20, [(i + 1, i) for i in range(9)] + [(19, 10), (19, 11), (18, 12), (19, 18), (15, 13), (19, 15), (10, 9)])))
    return out


@pytest.mark.parametrize("name,p", structures(), ids=[n for n, _ in structures()])
def test_schedule_switches_are_race_free(name, p):
    for two_side in (0, 1, 2):
        for overlap in (0, 1, 2):
            for split_u1 in (0, 1, 4):
                for gate in (0, 2):
                    for flow in (0, -1, 3):
                        r = chk(p, two_side=two_side, overlap=overlap, split_u1=split_u1, flood_gate=gate, factor_flow=flow, factor_flow_rows=64)
                        assert r["violations"] == 0, (name, two_side, overlap, split_u1, gate, flow, r)



def test_distributed_cuts_are_race_free_in_both_phases():
    nt = 48
    p = lower(nt, [(i, j) for i in range(nt) for j in range(max(0, i - 3), i)])
    perm = None
    for world in (2, 4):
        for rank in range(world):
            for flow in (0, -1, 4):
                r = chk(p, world=world, rank=rank, two_side=2, overlap=1, split_u1=1, flood_gate=0, factor_flow=flow, factor_flow_rows=64)
                assert r["violations"] == 0, (world, rank, flow, r)


def test_dataflow_launch_takes_the_top_of_a_dense_block():
    p = lower(18, [(i, j) for i in range(18) for j in range(i)])
    r = chk(p, factor_flow=64, factor_flow_rows=1000)
    # 18 potrf + 9 units per panel solve (153); of the 969 updates (sum of m (m + 1) / 2, m = 1..17) the 153 into the NEXT column
    # (the critical chain) keep nine 48 x 48 units each, the 816 = C(18, 3) whose target lies two columns or more ahead are one
    # whole-tile unit each (round 5): the whole factorisation
    assert r["violations"] == 0 and r["flow_groups"] == 18 and r["flow_units"] == 18 + 9 * (153 + 153) + 816, r
    auto = chk(p)
    assert auto["violations"] == 0 and auto["flow_groups"] >= 10, auto
