"""tests/golden/c3_quads.npz (tests/golden/make_golden_c3.py): QUADRILATERALS as committed records -- a room of 150 rectangles, 260 general
convex / tilted / coincident quadrilaterals and 120 triangles; rays aimed at corners, edge points, the diagonal both triangles of
Quadrilateral.Intersect share (Hare_Geometry_Polygons.cs:731-823), and interiors.  CPU: the oracle still produces them and the product's
host single-ray path equals them; GPU: every batch kernel equals them -- the voxel kernels with the quadrilateral pre-cull of round 5,
the octree and kd-tree kernels each forced in turn.  The same records are what bindings/csharp/tests/GoldenParity.cs compares the REFERENCE
classes with (case 3), for whoever has a .NET SDK."""
import os

import numpy as np
import pytest

import hare_amd as H
from oracle import pyoracle as po
from tests.helpers import assert_events_equal
from tests.test_shoot_one import shoot_all

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "c3_quads.npz"))
D, OD, OP, KDD, KDP = (int(x) for x in G["params"])
V, NV = np.ascontiguousarray(G["verts"]), np.ascontiguousarray(G["nverts"])


def test_the_set_really_is_about_quadrilaterals():
    assert int((NV == 4).sum()) >= 400 and int((NV == 3).sum()) >= 100
    hit = G["voxel"]["hit"] != 0
    assert hit.mean() > 0.99 and (NV[G["voxel"]["poly_id"][hit]] == 4).mean() > 0.8
    # general quadrilaterals: at least a hundred that are NOT parallelograms (v3 != v0 + v2 - v1)
    q = V[NV == 4]
    assert int((np.abs(q[:, 3] - (q[:, 0] + q[:, 2] - q[:, 1])).max(1) > 1e-9).sum()) >= 100
    # rays that hit in the second triangle (2,3,0) exist: u, v of the full test come from whichever triangle accepted
    assert (G["octree"]["hit"] != 0).sum() > 5000


def test_the_oracle_still_produces_the_committed_records():
    T = po.Topology(V, NV)
    rays, e1 = G["rays"], G["excl1"]
    vox = po.VoxelGrid([T], domain=D, build_mode=0)
    assert_events_equal(vox.shoot(rays)[0], G["voxel"], what="voxel")
    assert_events_equal(vox.shoot(rays, excl1=e1)[0], G["voxel_excl"], what="voxel excl")
    oc = po.Octree([T], OD, OP)
    assert_events_equal(oc.shoot(rays)[0], G["octree"], what="octree")
    assert_events_equal(oc.shoot(rays, excl1=e1)[0], G["octree_excl"], what="octree excl")
    assert_events_equal(po.KDTree([T], KDD, KDP).shoot(rays)[0], G["kdtree"], what="kdtree")


def test_host_single_ray_path_equals_the_committed_records(monkeypatch):
    monkeypatch.setenv("HARE_BUILD", "host")
    T = H.Topology(V, NV)
    k = slice(0, 2500)
    rays, e1 = G["rays"][k], G["excl1"][k]
    g = H.Voxel_Grid([T], D)
    assert_events_equal(shoot_all(g, rays)[0], G["voxel"][k], what="host voxel")
    assert_events_equal(shoot_all(g, rays, e1=e1)[0], G["voxel_excl"][k], what="host voxel excl")
    oc = H.Octree([T], OD, OP)
    assert_events_equal(shoot_all(oc, rays)[0], G["octree"][k], what="host octree")
    assert_events_equal(shoot_all(oc, rays, e1=e1)[0], G["octree_excl"][k], what="host octree excl")
    assert_events_equal(shoot_all(H.KDTree([T], KDD, KDP), rays[:800])[0], G["kdtree"][:800], what="host kd")


@pytest.mark.gpu
def test_batch_kernels_equal_the_committed_records():
    T = H.Topology(V, NV)
    rays, e1 = G["rays"], G["excl1"]
    g = H.Voxel_Grid([T], D)
    for kern, name in ((2, "hare_voxel_pool_quad"), (1, "hare_voxel_persist_quad")):
        g.set_option("voxel_kernel", kern)
        assert g.kernel_name(len(rays)) == name
        assert_events_equal(g.Shoot_batch(rays)[0], G["voxel"], what=f"gpu {name}")
        assert_events_equal(g.Shoot_batch(rays, poly_origin1=e1)[0], G["voxel_excl"], what=f"gpu {name} excl")
    assert_events_equal(g.Shoot_batch(rays, simple_kernel=True)[0], G["voxel"], what="gpu voxel simple")
    oc = H.Octree([T], OD, OP)
    for kern in (4, 3, 1):                                   # K2d, K2g, K2p
        oc.set_option("octree_kernel", kern)
        assert_events_equal(oc.Shoot_batch(rays)[0], G["octree"], what=f"gpu octree kernel {kern}")
        assert_events_equal(oc.Shoot_batch(rays, poly_origin1=e1)[0], G["octree_excl"], what=f"gpu octree kernel {kern} excl")
    kd = H.KDTree([T], KDD, KDP)
    for kern in (2, 1):                                      # K3d, one ray per lane
        kd.set_option("kdtree_kernel", kern)
        assert_events_equal(kd.Shoot_batch(rays)[0], G["kdtree"], what=f"gpu kd kernel {kern}")
