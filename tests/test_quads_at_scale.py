"""Quadrilaterals at BASELINE scale: `hall_quads` -- the 100k-triangle hall with its flat lattices left
un-split, 39 263 planar quadrilaterals + 22 382 triangles -- through Voxel_Grid D = 64, Octree 8 / 16 and KDTree, against the oracle's
restatement of Quadrilateral.Intersect (Hare_Geometry_Polygons.cs:784-823 fast variant for Voxel_Grid, :731-782 full variant with u, v for
the trees).  Until round 5 the quad paths had only met soups of <= 1 000 polygons."""
import numpy as np
import pytest

import hare_amd as H
from oracle import pyoracle as po
from tests.helpers import assert_events_equal

KD = (16, 8)


@pytest.fixture(scope="module")
def scene():
    m = H.scenes.hall_quads()
    assert m.P == 61645 and int((m.nverts == 4).sum()) == 39263
    return m, H.Topology(m.verts, m.nverts), po.Topology(m.verts, m.nverts)


def test_hall_quads_covers_the_hall_with_planar_quadrilaterals():
    q, t = H.scenes.hall_quads(), H.scenes.hall()
    v4 = q.verts[q.nverts == 4]
    nrm = np.cross(v4[:, 1] - v4[:, 0], v4[:, 2] - v4[:, 0])
    assert np.abs(np.einsum("ij,ij->i", nrm, v4[:, 3] - v4[:, 0])).max() == 0.0          # exactly planar on the 2^-8 m lattice

    def area(v):
        return 0.5 * np.linalg.norm(np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]), axis=1).sum()
    assert area(q.verts[q.nverts == 3]) + area(v4[:, [0, 1, 2]]) + area(v4[:, [0, 2, 3]]) == pytest.approx(area(t.verts), rel=1e-12)


def test_single_ray_host_path_on_the_quad_hall(scene, monkeypatch):
    """hare_shoot_one (the product's host trace) on 1 500 burst rays, all three partitions, against the oracle: no GPU."""
    monkeypatch.setenv("HARE_BUILD", "host")
    m, T, To = scene
    rays = H.scenes.burst_rays(1500, m.size)
    for part, orc in ((H.Voxel_Grid([T], 32), po.VoxelGrid([To], domain=32)), (H.Octree([T], 6, 16), po.Octree([To], 6, 16)),
                      (H.KDTree([T], 10, 16), po.KDTree([To], 10, 16))):
        ref, _ = orc.shoot(rays, nthreads=4)
        got = np.zeros(len(rays), H.capi.XEVENT_DTYPE)
        for i in range(len(rays)):
            got[i] = part.Shoot_one(rays[i].copy())
        assert_events_equal(got, ref, what=type(part).__name__)
        assert ref["hit"].sum() > 1400
        hit_quads = m.nverts[ref["poly_id"][ref["hit"] != 0]] == 4
        assert hit_quads.mean() > 0.5                                   # most of what a burst ray meets here IS a quadrilateral


@pytest.mark.gpu
def test_one_million_burst_rays_into_the_quad_hall_voxel_and_octree(scene):
    m, T, To = scene
    n = 1 << 20
    rays = H.scenes.burst_rays(n, m.size)
    g, og = H.Voxel_Grid([T], 64), po.VoxelGrid([To], domain=64)
    assert g.kernel_name(n) == "hare_voxel_pool_quad"
    ref, rc = og.shoot(rays, nthreads=16)
    ev, c = g.Shoot_batch(rays)
    assert_events_equal(ev, ref, what="voxel, 1M rays, quads")
    assert c["hits"] == rc["hits"] == n                                 # a closed room
    e1 = ref["poly_id"].astype(np.int32)                               # the exclusion overload: every ray re-cast past the polygon it hit
    ref2, _ = og.shoot(rays, excl1=e1, nthreads=16)
    ev2, _ = g.Shoot_batch(rays, poly_origin1=e1)
    assert_events_equal(ev2, ref2, what="voxel, 1M rays, quads, exclusions")
    del g
    oc, oo = H.Octree([T], 8, 16), po.Octree([To], 8, 16)
    assert oc.kernel_name(n) == "hare_octree_dense"
    ref, rc = oo.shoot(rays, nthreads=16)
    ev, c = oc.Shoot_batch(rays)
    assert_events_equal(ev, ref, what="octree, 1M rays, quads")
    assert c["hits"] == rc["hits"]
    small = rays[:60_000]                                               # ... and the kernel small batches get (eight lanes per ray: below 320 rays per CU)
    assert oc.kernel_name(len(small)) == "hare_octree_group"
    assert_events_equal(oc.Shoot_batch(small)[0], ref[:60_000], what="octree, 60k rays, quads")
    mid = rays[:100_000]                                                # ... and K2d with half-full waves (round 6: from 81 920 rays)
    assert oc.kernel_name(len(mid)) == "hare_octree_dense"
    assert_events_equal(oc.Shoot_batch(mid)[0], ref[:100_000], what="octree, 100k rays, quads")


@pytest.mark.gpu
def test_kdtree_on_the_quad_hall(scene):
    """KDTree.Shoot visits every leaf in the reference (F4), so the oracle is O(P) per ray: 65 536 of the 1M burst rays are compared
    (every 16th: all polar bands), the GPU casts all of them."""
    m, T, To = scene
    n = 1 << 20
    rays = H.scenes.burst_rays(n, m.size)
    kd, ok = H.KDTree([T], *KD), po.KDTree([To], *KD)
    ev, c = kd.Shoot_batch(rays)
    sample = np.arange(0, n, 16)
    ref, _ = ok.shoot(rays[sample], nthreads=16)
    assert_events_equal(ev[sample], ref, what="kd-tree, 65 536 of 1M rays, quads")
    assert c["hits"] == n
