"""tests/golden/c2_ties.npz (tests/golden/make_golden_c2.py): exact ties and trees over two topologies, as committed records.
CPU: the oracle still produces them (a change of the oracle shows here) and the product's host single-ray path equals them;
GPU: the batch kernels equal them.  The same records are what bindings/csharp/tests/GoldenParity.cs compares the REFERENCE
classes with (case 2), for whoever has a .NET SDK."""
import os

import numpy as np
import pytest

import hare_amd as H
from oracle import pyoracle as po
from tests.helpers import assert_events_equal
from tests.test_shoot_one import shoot_all

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "c2_ties.npz"))
D, OD, OP, KDD, KDP = (int(x) for x in G["params"])


def models():
    def pad(v):
        out = np.zeros((len(v), 4, 3)); out[:, :3] = v
        return np.ascontiguousarray(out), np.full(len(v), 3, np.int32)
    return pad(G["verts0"]), pad(G["verts1"])


def test_the_oracle_still_produces_the_committed_records():
    (v0, n0), (v1, n1) = models()
    T0, T1 = po.Topology(v0, n0), po.Topology(v1, n1)
    rays, e1, e1b = G["rays"], G["excl1"], G["excl1_two"]
    vox = po.VoxelGrid([T0], domain=D, build_mode=0)
    assert_events_equal(vox.shoot(rays)[0], G["voxel"], what="voxel")
    assert_events_equal(vox.shoot(rays, excl1=e1)[0], G["voxel_excl"], what="voxel excl")
    assert_events_equal(po.Octree([T0], OD, OP).shoot(rays)[0], G["octree"], what="octree")
    assert_events_equal(po.KDTree([T0], KDD, KDP).shoot(rays)[0], G["kdtree"], what="kdtree")
    oc2, kd2 = po.Octree([T0, T1], OD, OP), po.KDTree([T0, T1], KDD, KDP)
    for top in (0, 1):
        assert_events_equal(oc2.shoot(rays, top_index=top)[0], G["octree2_top%d" % top], what="octree2 top %d" % top)
        assert_events_equal(oc2.shoot(rays, top_index=top, excl1=e1b)[0], G["octree2_top%d_excl" % top], what="octree2 excl top %d" % top)
        assert_events_equal(kd2.shoot(rays, top_index=top)[0], G["kdtree2_top%d" % top], what="kdtree2 top %d" % top)


def test_host_single_ray_path_equals_the_committed_records(monkeypatch):
    monkeypatch.setenv("HARE_BUILD", "host")
    (v0, n0), (v1, n1) = models()
    T0, T1 = H.Topology(v0, n0), H.Topology(v1, n1)
    k = slice(0, 2500)
    rays, e1, e1b = G["rays"][k], G["excl1"][k], G["excl1_two"][k]
    g = H.Voxel_Grid([T0], D)
    assert_events_equal(shoot_all(g, rays)[0], G["voxel"][k], what="host voxel")
    assert_events_equal(shoot_all(g, rays, e1=e1)[0], G["voxel_excl"][k], what="host voxel excl")
    assert_events_equal(shoot_all(H.Octree([T0], OD, OP), rays)[0], G["octree"][k], what="host octree")
    assert_events_equal(shoot_all(H.Octree([T0], OD, OP), rays, e1=e1)[0], G["octree_excl"][k], what="host octree excl")
    assert_events_equal(shoot_all(H.KDTree([T0], KDD, KDP), rays[:800])[0], G["kdtree"][:800], what="host kd")
    oc2, kd2 = H.Octree([T0, T1], OD, OP), H.KDTree([T0, T1], KDD, KDP)
    for top in (0, 1):
        assert_events_equal(shoot_all(oc2, rays, top=top)[0], G["octree2_top%d" % top][k], what="host octree2 top %d" % top)
        assert_events_equal(shoot_all(oc2, rays, e1=e1b, top=top)[0], G["octree2_top%d_excl" % top][k], what="host octree2 excl top %d" % top)
        assert_events_equal(shoot_all(kd2, rays[:800], top=top)[0], G["kdtree2_top%d" % top][:800], what="host kd2 top %d" % top)


@pytest.mark.gpu
def test_batch_kernels_equal_the_committed_records():
    (v0, n0), (v1, n1) = models()
    T0, T1 = H.Topology(v0, n0), H.Topology(v1, n1)
    rays, e1, e1b = G["rays"], G["excl1"], G["excl1_two"]
    g = H.Voxel_Grid([T0], D)
    assert_events_equal(g.Shoot_batch(rays)[0], G["voxel"], what="gpu voxel")
    assert_events_equal(g.Shoot_batch(rays, poly_origin1=e1)[0], G["voxel_excl"], what="gpu voxel excl")
    assert_events_equal(H.Octree([T0], OD, OP).Shoot_batch(rays)[0], G["octree"], what="gpu octree")
    assert_events_equal(H.Octree([T0], OD, OP).Shoot_batch(rays, poly_origin1=e1)[0], G["octree_excl"], what="gpu octree excl")
    assert_events_equal(H.KDTree([T0], KDD, KDP).Shoot_batch(rays)[0], G["kdtree"], what="gpu kd")
    oc2, kd2 = H.Octree([T0, T1], OD, OP), H.KDTree([T0, T1], KDD, KDP)
    for top in (0, 1):
        assert_events_equal(oc2.Shoot_batch(rays, top_index=top)[0], G["octree2_top%d" % top], what="gpu octree2 top %d" % top)
        assert_events_equal(oc2.Shoot_batch(rays, top_index=top, poly_origin1=e1b)[0], G["octree2_top%d_excl" % top], what="gpu octree2 excl")
        assert_events_equal(kd2.Shoot_batch(rays, top_index=top)[0], G["kdtree2_top%d" % top], what="gpu kd2 top %d" % top)
