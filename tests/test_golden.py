"""Committed golden vectors for BASELINE.json config 1 (10k random rays + edge cases into the 1k-tri
shoebox).  CPU: the oracle reproduces them (regression pin).  GPU: the HIP path reproduces them
through the C-ABI, bit for bit."""
import os

import numpy as np
import pytest

import hare_amd as H
from oracle import pyoracle as po
from tests.helpers import assert_events_equal

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "c1_shoebox.npz"))
D, OD, OP, KDD, KDP = (int(x) for x in G["params"])


def test_inputs_are_reproducible_from_the_seeded_generators():
    m = H.scenes.shoebox()
    assert m.P == 972
    assert np.array_equal(G["rays"][:10000], H.scenes.random_rays(10000, m.size))


def test_oracle_reproduces_golden_vectors():
    m = H.scenes.shoebox()
    T = po.Topology(m.verts, m.nverts)
    rays, e1 = G["rays"], G["excl1"]
    vox = po.VoxelGrid([T], domain=D)
    ev, ctr = vox.shoot(rays)
    assert_events_equal(ev, G["voxel"], what="voxel")
    assert [ctr[k] for k in ("cells", "entries", "tests")] == list(G["voxel_ctr"])
    assert_events_equal(vox.shoot(rays, excl1=e1)[0], G["voxel_excl"], what="voxel excl")
    oc = po.Octree([T], OD, OP)
    assert_events_equal(oc.shoot(rays)[0], G["octree"], what="octree")
    assert_events_equal(oc.shoot(rays, excl1=e1)[0], G["octree_excl"], what="octree excl")
    assert_events_equal(po.KDTree([T], KDD, KDP).shoot(rays)[0], G["kdtree"], what="kdtree")


def test_golden_sanity():
    v = G["voxel"]
    assert v[:10000]["hit"].all()                     # closed room, origins inside
    assert (G["voxel_excl"]["poly_id"] != G["excl1"])[G["voxel_excl"]["hit"] == 1].all()
    nanrows = np.isnan(G["rays"]).any(axis=1)
    assert nanrows.sum() == 2 and not v["hit"][np.isnan(G["rays"][:, 0])].any()


@pytest.mark.gpu
def test_hip_voxel_reproduces_golden_vectors():
    m = H.scenes.shoebox()
    g = H.Voxel_Grid([H.Topology(m.verts, m.nverts)], D)
    ev, ctr = g.Shoot_batch(G["rays"], count_work=True)
    assert_events_equal(ev, G["voxel"], what="hip voxel")
    assert ctr["cells"] == int(G["voxel_ctr"][0]) and ctr["entries"] == int(G["voxel_ctr"][1])
    assert ctr["hits"] == int(G["voxel"]["hit"].sum()) and ctr["rays"] == len(ev)
    ev, _ = g.Shoot_batch(G["rays"], poly_origin1=G["excl1"])
    assert_events_equal(ev, G["voxel_excl"], what="hip voxel excl")


@pytest.mark.gpu
def test_hip_octree_reproduces_golden_vectors():
    m = H.scenes.shoebox()
    g = H.Octree([H.Topology(m.verts, m.nverts)], OD, OP)
    assert_events_equal(g.Shoot_batch(G["rays"])[0], G["octree"], what="hip octree")
    assert_events_equal(g.Shoot_batch(G["rays"], poly_origin1=G["excl1"])[0], G["octree_excl"], what="hip octree excl")


@pytest.mark.gpu
def test_hip_kdtree_reproduces_golden_vectors():
    m = H.scenes.shoebox()
    g = H.KDTree([H.Topology(m.verts, m.nverts)], KDD, KDP)
    assert_events_equal(g.Shoot_batch(G["rays"])[0], G["kdtree"], what="hip kdtree")
