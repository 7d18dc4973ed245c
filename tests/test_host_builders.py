"""The product's partition builders (hare_amd/csrc/build_host.cpp) against the oracle's restatement
of the reference constructors: identical candidate lists, node boxes and ordering.  Order decides
exact-t ties, so equality is required, not equivalence.  CPU only."""
import numpy as np
import pytest

import hare_amd as H
from oracle import pyoracle as po
from tests.helpers import soup


def topo_pair(verts, nverts):
    return H.Topology(verts, nverts), po.Topology(verts, nverts)


@pytest.fixture(scope="module")
def scenes3():
    m = H.scenes.shoebox()
    v, nv, _ = soup()
    hall = H.scenes.hall(edge=1.0)   # ~11k triangles: displaced ceiling, solids
    return [("shoebox", m.verts, m.nverts), ("soup", v, nv), ("hall-coarse", hall.verts, hall.nverts)]


def test_normals_and_bounds_match_polygon_ctor_and_finish_topology(scenes3):
    for name, v, nv in scenes3:
        a, b = topo_pair(v, nv)
        assert np.array_equal(a.normals, b.normals), name
        assert np.array_equal(np.signbit(a.normals), np.signbit(b.normals)), name
        assert np.array_equal(a.Min, b.min) and np.array_equal(a.Max, b.max), name


@pytest.mark.parametrize("domain", [1, 5, 8, 16])
def test_fixed_voxel_lists_identical(scenes3, domain):
    for name, v, nv in scenes3:
        a, b = topo_pair(v, nv)
        g = H.Voxel_Grid([a], domain)
        # the oracle's literal D^3 x P loop is the reference's own algorithm; use it where it is cheap
        o = po.VoxelGrid([b], domain=domain, build_mode=0 if (domain ** 3) * b.P < 3e7 else 1)
        s, i = g.Voxel_Inv()
        so, io = o.lists()
        assert np.array_equal(s, so) and np.array_equal(i, io), (name, domain)
        info = g.info()
        assert info.ct == o.ct and g.Char_Step == o.char_step
        assert tuple(info.obox_min) == tuple(o.obox_min) and tuple(info.obox_max) == tuple(o.obox_max)
        assert tuple(info.voxel_dims) == tuple(o.voxel_dims)
        assert (g.Xdim, g.Ydim, g.Zdim) == tuple(o.obox_max - o.obox_min)


def test_adaptive_voxel_lists_identical(scenes3):
    for name, v, nv in scenes3:
        a, b = topo_pair(v, nv)
        for max_domain, avg in ((3, 4), (5, 10), (6, 40)):
            g = H.Voxel_Grid([a], max_domain, avg)
            o = po.VoxelGrid([b], max_domain=max_domain, avg_polys=avg)
            assert g.VoxelCt == o.ct, (name, max_domain, avg)
            s, i = g.Voxel_Inv()
            so, io = o.lists()
            assert np.array_equal(s, so) and np.array_equal(i, io), (name, max_domain, avg)


def test_two_topologies_share_one_grid():
    m = H.scenes.shoebox()
    v, nv, _ = soup(100, 30)
    a0, b0 = topo_pair(m.verts, m.nverts)
    a1, b1 = topo_pair(v, nv)
    g = H.Voxel_Grid([a0, a1], 6)
    o = po.VoxelGrid([b0, b1], domain=6, build_mode=0)
    for top in (0, 1):
        s, i = g.Voxel_Inv(top)
        so, io = o.lists(top)
        assert np.array_equal(s, so) and np.array_equal(i, io)


def test_octree_nodes_identical(scenes3):
    for name, v, nv in scenes3:
        a, b = topo_pair(v, nv)
        for depth, mp in ((3, 4), (5, 16)):
            g = H.Octree([a], depth, mp)
            o = po.Octree([b], depth, mp)
            for x, y in zip(g.nodes(), o.export()):
                assert np.array_equal(x, y), (name, depth, mp)


def test_kdtree_nodes_identical(scenes3):
    for name, v, nv in scenes3:
        a, b = topo_pair(v, nv)
        for depth, mp in ((4, 8), (10, 4)):
            g = H.KDTree([a], depth, mp)
            o = po.KDTree([b], depth, mp)
            for x, y in zip(g.nodes(), o.export()):
                assert np.array_equal(x, y), (name, depth, mp)


def test_public_members_of_voxel_grid():
    m = H.scenes.shoebox()
    g = H.Voxel_Grid([H.Topology(m.verts, m.nverts)], 8)
    info = g.info()
    assert g.MinPt == tuple(info.obox_min)
    x, y, z = g.PointInVoxel((5.0, 3.5, 2.0))
    assert (x, y, z) == tuple(int(np.floor((p - info.obox_min[a]) / info.voxel_dims[a])) for a, p in enumerate((5.0, 3.5, 2.0)))
    assert g.VoxelCode(1, 2, 3) == 64 * 3 + 8 * 1 + 2      # XYTot*Z + VoxelCtY*X + Y (Voxel_Grid.cs:264-267)


def test_scene_options_are_range_checked_without_a_gpu():
    """hare_scene_set_option (include/hare_hip.h): the A/B switches the GPU tests and tools use exist on a scene built on the host,
    refuse values outside their range and refuse unknown names -- nothing here launches a kernel."""
    m = H.scenes.shoebox()
    g = H.Voxel_Grid([H.Topology(m.verts, m.nverts)], 8)
    for name, good, bad in (("voxel_kernel", 2, 3), ("octree_kernel", 4, 5), ("octree_tail", 1, 3), ("coop_tail", 0, 2), ("wide_drain", 0, 2), ("ticket_rays", 64, 1 << 20)):
        g.set_option(name, good)
        with pytest.raises(H.HareError):
            g.set_option(name, bad)
        g.set_option(name, {"coop_tail": 1, "wide_drain": 1, "octree_tail": 2}.get(name, 0))
    with pytest.raises(H.HareError):
        g.set_option("no_such_option", 1)
    # hare_scene_get_option: every option reads back; the memory figures are 0 on a scene that never saw a device
    g.set_option("ticket_rays", 96)
    assert g.get_option("ticket_rays") == 96 and g.get_option("voxel_tight") == 1 and g.get_option("voxel_tight_max_mb") == 0
    assert g.get_option("voxel_tight_bytes") == 0 and g.get_option("octree_scratch_bytes") == 0
    with pytest.raises(H.HareError):
        g.get_option("no_such_option")


def test_octree_child_boxes_follow_from_the_parent_box():
    """The persistent octree kernel does not load child boxes: it derives the children's planes from the
    parent's stored box with BuildOctree's expressions ("Octree - alt.cs":96-111).  Check on the built tree
    that this derivation reproduces every stored child box bit for bit."""
    m = H.scenes.hall(edge=1.0)
    g = H.Octree([H.Topology(m.verts, m.nverts)], 6, 8)
    boxes, fc, _, _, _ = g.nodes()
    interior = np.nonzero(fc >= 0)[0]
    assert len(interior) > 100
    for n in interior:
        mn, mx = boxes[n, :3], boxes[n, 3:]
        center = (mx + mn) / 2
        for i in range(8):
            bits = [(i & 4) != 0, (i & 2) != 0, (i & 1) != 0]
            cmin = np.array([(center[a] if bits[a] else mn[a]) - 0.1 for a in range(3)])
            cmax = np.array([(mx[a] if bits[a] else center[a]) + 0.1 for a in range(3)])
            child = boxes[fc[n] + i]
            assert np.array_equal(child[:3], cmin) and np.array_equal(child[3:], cmax), (n, i)
