"""Host-only checks of what apexgpu_set_structure derives from the observation list before it touches the device
(csrc/ba_structure.h, through apexgpu_debug_host_structure): the internal camera order is a permutation, a banded
capture keeps every camera in the dissection, hub cameras and accidental long-range matches are moved to a dense
border (and the fill shrinks accordingly), tree sharding gives every landmark to exactly one rank.  No GPU."""
import numpy as np
import pytest

import apex_solver_amd as pkg
from apex_solver_amd import capi


def test_banded_capture_has_no_border_and_a_bushy_tree():
    d = pkg.synthetic.make_named("final-13682", 0.05)
    st = capi.host_structure(d.n_cam, d.n_pt, d.cam_idx, d.pt_idx)
    assert sorted(st["cmap"].tolist()) == list(range(d.n_cam))
    assert st["hub_cameras"] == 0 and st["border_tiles"] == 1
    nat = capi.host_structure(d.n_cam, d.n_pt, d.cam_idx, d.pt_idx, nested_dissection=0)
    assert st["etree_levels"] < nat["etree_levels"]                      # the chain became a tree
    assert st["touched_tiles"] == nat["touched_tiles"]


def test_hub_cameras_and_long_range_matches_go_to_the_border():
    d = pkg.synthetic.make_named("final-13682-hub", 0.2)
    n_hub = max(1, int(round(0.015 * d.n_cam)))
    hubs = (np.arange(n_hub) * d.n_cam) // n_hub + (d.n_cam // (2 * n_hub))
    st = capi.host_structure(d.n_cam, d.n_pt, d.cam_idx, d.pt_idx)
    off = capi.host_structure(d.n_cam, d.n_pt, d.cam_idx, d.pt_idx, hubs_last=0)
    assert sorted(st["cmap"].tolist()) == list(range(d.n_cam))
    nb = int(st["hub_cameras"])
    print("border cameras", nb, "of", d.n_cam, "tiles", st["tiles"], "vs", off["tiles"], "levels", st["etree_levels"], "vs", off["etree_levels"])
    assert n_hub <= nb <= d.n_cam // 8
    assert (st["cmap"][hubs] >= d.n_cam - nb).all()                      # every generated hub sits in the border
    assert st["tiles"] < 0.5 * off["tiles"] and st["etree_levels"] < 0.5 * off["etree_levels"]
    assert off["hub_cameras"] == 0


@pytest.mark.parametrize("world", [2, 4])
def test_tree_sharding_partitions_the_landmarks(world):
    d = pkg.synthetic.make_problem(1500, 30000, 3, 7, config_id=310)
    owned = []
    for r in range(world):
        st = capi.host_structure(d.n_cam, d.n_pt, d.cam_idx, d.pt_idx, rank=r, world=world)
        assert st["tree_sharded"] == 1.0 and st["top_columns"] > 0
        owned.append(st["owned"])
    owned = np.stack(owned)
    assert (owned.sum(0) == 1).all()
    share = owned.sum(1) / d.n_pt
    assert share.max() < 1.6 / world
    # range sharding: contiguous, balanced by observations
    st = capi.host_structure(d.n_cam, d.n_pt, d.cam_idx, d.pt_idx, rank=0, world=world, tree_sharding=0)
    assert st["tree_sharded"] == 0.0 and st["owned"][: int(st["owned"].sum())].all()


def test_observation_order_does_not_change_the_structure():
    """Observations grouped by landmark (BAL files, the synthetic generator: the set-up's fast path) or in any order:
    same camera order, same tile structure, same pair count."""
    d = pkg.synthetic.make_problem(700, 20000, 3, 8, config_id=77)
    a = capi.host_structure(d.n_cam, d.n_pt, d.cam_idx, d.pt_idx)
    perm = np.random.default_rng(3).permutation(d.n_obs)
    b = capi.host_structure(d.n_cam, d.n_pt, d.cam_idx[perm], d.pt_idx[perm])
    assert np.array_equal(a["cmap"], b["cmap"])
    for k in ("tiles", "touched_tiles", "etree_levels", "pair_contributions", "pair_blocks", "hub_cameras"):
        assert a[k] == b[k], k
