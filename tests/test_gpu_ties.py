"""Exact ties, where bit-exactness is decided by comparison operators rather than by arithmetic: rays aimed EXACTLY (lattice
coordinates, exact FP64 differences) at vertices, edge midpoints and points of edges shared by neighbouring polygons; coincident
polygons (the same corners twice, also in another corner order), coplanar overlapping polygons (equal t: the strict `<` of
Voxel_Grid.cs:707 / "Octree - alt.cs":233 keeps the first in list order).  Barycentric coordinates are exactly 0 or 1 on such
hits, which is also where a conservative FP32 pre-cull has the least room.  Every kernel, the host single-ray path and both
builders against the oracle."""
import numpy as np
import pytest

import hare_amd as H
from oracle import pyoracle as po
from tests.helpers import assert_events_equal
from tests.test_shoot_one import shoot_all

LAT = H.scenes.LATTICE


def tie_scene(seed=0):
    """A lattice-snapped shoebox (shared edges everywhere) + a soup of lattice triangles, 25 % of the polygons once more at the
    end (half of them with rotated corner order), + coplanar overlapping pairs on the box's floor."""
    rng = np.random.default_rng(seed)
    m = H.scenes.shoebox(nface=6, size=(8.0, 6.0, 4.0))
    V = [np.asarray(m.verts, np.float64).reshape(-1, 4, 3).copy()]
    NV = [np.asarray(m.nverts, np.int32).copy()]
    P = 160
    c = H.scenes.snap(rng.uniform(1.0, 3.0, (P, 3)) * [2.0, 1.5, 1.0])
    a = H.scenes.snap(rng.uniform(-0.75, 0.75, (P, 3)))
    b = H.scenes.snap(rng.uniform(-0.75, 0.75, (P, 3)))
    tri = np.zeros((P, 4, 3)); tri[:, 0] = c; tri[:, 1] = c + a; tri[:, 2] = c + b
    ok = np.linalg.norm(np.cross(a, b), axis=1) > 1e-3
    V.append(tri[ok]); NV.append(np.full(int(ok.sum()), 3, np.int32))
    # coplanar overlapping pairs lying ON the floor z = 0 (same plane as the box's floor triangles: equal t three ways)
    q = H.scenes.snap(rng.uniform(1.0, 5.0, (12, 2)))
    for k in range(12):
        x, y = q[k]
        V.append(np.array([[[x, y, 0], [x + 1.5, y, 0], [x, y + 1.25, 0], [0, 0, 0]],
                           [[x + 0.5, y + 0.25, 0], [x + 2.0, y + 0.25, 0], [x + 0.5, y + 1.5, 0], [0, 0, 0]]], np.float64))
        NV.append(np.full(2, 3, np.int32))
    verts = np.concatenate(V); nverts = np.concatenate(NV)
    dup = rng.choice(len(nverts), len(nverts) // 4, replace=False)
    dv = verts[dup].copy()
    rot = np.arange(len(dup)) % 2 == 1
    tri_rot = rot & (nverts[dup] == 3)
    dv[tri_rot, :3] = dv[tri_rot][:, [1, 2, 0]]                     # same triangle, corners rotated
    verts = np.concatenate([verts, dv]); nverts = np.concatenate([nverts, nverts[dup]])
    return np.ascontiguousarray(verts), np.ascontiguousarray(nverts), (8.0, 6.0, 4.0)


def tie_rays(verts, nverts, size, n=6000, seed=1):
    """Origins on the lattice inside the box; targets: polygon corners, edge midpoints, quarter points of edges, centroids-on-
    lattice; direction = target - origin, exact.  A third of the rays are not normalised further, the rest are scaled by powers
    of two (exact), some start ON a polygon's plane or on a voxel face."""
    rng = np.random.default_rng(seed)
    P = len(nverts)
    o = H.scenes.snap(rng.uniform(0.25, 0.75, (n, 3)) * np.asarray(size))
    o[::17, 2] = 0.0                                                 # on the floor plane (coplanar with floor polygons)
    o[5::23, 0] = H.scenes.snap(rng.integers(1, 8, len(o[5::23])) * (size[0] / 8.0))   # on voxel faces of an 8-cell grid
    p = rng.integers(0, P, n)
    k0 = rng.integers(0, 3, n); k1 = (k0 + 1) % 3
    A, B = verts[p, k0], verts[p, k1]
    kind = rng.integers(0, 4, n)
    tgt = np.where((kind == 0)[:, None], A, np.where((kind == 1)[:, None], (A + B) / 2, np.where((kind == 2)[:, None], A + (B - A) / 4,
                   (verts[p, 0] + verts[p, 1]) / 2 + (verts[p, 2] - verts[p, 0]) / 4)))
    d = tgt - o
    d[np.all(d == 0, axis=1)] = (1.0, 0.0, 0.0)
    s = np.where(rng.random(n) < 0.33, 1.0, 2.0 ** rng.integers(-3, 4, n).astype(np.float64))
    return np.ascontiguousarray(np.concatenate([o, d * s[:, None]], 1))


@pytest.fixture(scope="module")
def ties():
    v, nv, size = tie_scene()
    return v, nv, size, tie_rays(v, nv, size)


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["persist", "pool"])
@pytest.mark.parametrize("domain", [1, 8, 21])
def test_voxel_exact_ties(ties, kernel, domain, monkeypatch):
    monkeypatch.setenv("HARE_DEV", "1")     # developer overrides are read (once, at scene creation) only in a process that opted in
    monkeypatch.setenv("HARE_VOXEL_KERNEL", kernel)
    v, nv, size, rays = ties
    T, To = H.Topology(v, nv), po.Topology(v, nv)
    g, o = H.Voxel_Grid([T], domain), po.VoxelGrid([To], domain=domain)
    ref, rc = o.shoot(rays)
    ev, c = g.Shoot_batch(rays)
    assert_events_equal(ev, ref, what=f"ties voxel {kernel} D={domain}")
    assert c["hits"] == rc["hits"] and 0 < rc["hits"] <= len(rays)
    e1 = ref["poly_id"].astype(np.int32)                            # leave the polygon just hit: its twin answers
    assert_events_equal(g.Shoot_batch(rays, poly_origin1=e1)[0], o.shoot(rays, excl1=e1)[0], what=f"ties voxel {kernel} D={domain} excl")
    assert_events_equal(g.Shoot_batch(rays, simple_kernel=True)[0], ref, what=f"ties voxel simple D={domain}")
    if kernel == "persist":
        assert_events_equal(shoot_all(g, rays[:1500])[0], ref[:1500], what=f"ties voxel host D={domain}")


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["group", "dense", "persist", "pool"])
def test_tree_exact_ties(ties, kernel, monkeypatch):
    monkeypatch.setenv("HARE_DEV", "1")     # developer overrides are read (once, at scene creation) only in a process that opted in
    monkeypatch.setenv("HARE_OCTREE_KERNEL", kernel)
    v, nv, size, rays = ties
    T, To = H.Topology(v, nv), po.Topology(v, nv)
    for depth, maxp in ((4, 8), (7, 2)):
        oc, oo = H.Octree([T], depth, maxp), po.Octree([To], depth, maxp)
        ref, _ = oo.shoot(rays)
        assert_events_equal(oc.Shoot_batch(rays)[0], ref, what=f"ties octree {kernel} {depth}/{maxp}")
        e1 = ref["poly_id"].astype(np.int32)
        assert_events_equal(oc.Shoot_batch(rays, poly_origin1=e1)[0], oo.shoot(rays, excl1=e1)[0], what=f"ties octree {kernel} excl")
        if kernel == "persist":
            assert_events_equal(oc.Shoot_batch(rays, simple_kernel=True)[0], ref, what="ties octree simple")
            assert_events_equal(shoot_all(oc, rays[:1500])[0], ref[:1500], what="ties octree host")
    if kernel == "persist":
        kd, ko = H.KDTree([T], 7, 6), po.KDTree([To], 7, 6)
        assert_events_equal(kd.Shoot_batch(rays[:2500])[0], ko.shoot(rays[:2500])[0], what="ties kd")


def test_tie_scene_really_has_ties():
    """The scene does what its docstring says (CPU, oracle only): many rays hit exactly on an edge or a corner, and a
    coincident twin exists for a quarter of the polygons -- otherwise the GPU tests above would be ordinary parity tests."""
    v, nv, size = tie_scene()
    rays = tie_rays(v, nv, size, n=3000)
    ref, rc = po.VoxelGrid([po.Topology(v, nv)], domain=8).shoot(rays)
    hit = ref["hit"] != 0
    assert hit.sum() > 2000
    # a ray aimed at a lattice point of a polygon reaches that point exactly or is stopped earlier; count exact arrivals
    o, d = rays[:, :3], rays[:, 3:]
    at_target = hit & (np.abs(ref["t"] - 1.0) < 1e-12)                 # direction = target - origin (x a power of two): t = 1 / scale
    scaled = hit & np.isin(ref["t"], 2.0 ** np.arange(-4.0, 5.0))
    assert (at_target | scaled).sum() > 500
    keys = {}
    for p in range(len(nv)):
        k = tuple(sorted(map(tuple, v[p, :nv[p]])))
        keys.setdefault(k, []).append(p)
    assert sum(1 for k in keys.values() if len(k) > 1) >= len(nv) // 6
