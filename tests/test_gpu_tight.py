"""The subtree TIGHT BOXES of the octree kernels K2d / K2p / K2g (api.cpp: make_tight_boxes; scene option `octree_tight`): a popped node whose
subtree's polygons the ray cannot hit is dropped without being visited.  Not in the reference -- "Octree - alt.cs":207-237 visits every
node its loose boxes let through and lets RayXtri say no -- so the only acceptable effect is none: the same eight X_Event fields, bit for
bit, with the boxes on, off, and from the oracle.  The cases are chosen where a box test has the least room: rays aimed exactly at
polygon corners (the corners of the polygons' boxes), rays lying IN the faces of those boxes (a direction component exactly zero and the
origin on a vertex coordinate), rays inside a polygon's plane, coincident polygons, quadrilaterals, origins far outside the scene (the
boxes' margin is sized for origins within 1 024 extents: beyond that the kernels must not use them), two topologies, deep trees."""
import numpy as np
import pytest

import hare_amd as H
from oracle import pyoracle as po
from tests.helpers import assert_events_equal, soup, soup_rays
from tests.test_gpu_round2 import deep_scene
from tests.test_gpu_ties import tie_rays, tie_scene

pytestmark = pytest.mark.gpu
KERNELS = {"dense": 4, "persist": 1, "group": 3}


def both_ways(oc, oo, rays, what, **kw):
    """HIP with the boxes on and off, per kernel, against the oracle."""
    okw = {("excl1" if k == "poly_origin1" else "excl2"): v for k, v in kw.items()}
    ref, rc = oo.shoot(rays, **okw)
    for name, k in KERNELS.items():
        oc.set_option("octree_kernel", k)
        for tight in (1, 0):
            oc.set_option("octree_tight", tight)
            ev, c = oc.Shoot_batch(rays, **kw)
            assert_events_equal(ev, ref, what=f"{what} {name} tight={tight}")
            assert c["hits"] == rc["hits"]
    oc.set_option("octree_kernel", 0)
    oc.set_option("octree_tight", 1)
    return ref


def face_rays(verts, nverts, size, n=4000, seed=2):
    """Rays that lie in the faces / edges of polygon bounding boxes: the origin takes one or two coordinates from a polygon corner and
    the direction is exactly zero along them (so the ray never leaves that plane / line), the rest is random; and rays that start
    exactly ON a corner."""
    rng = np.random.default_rng(seed)
    P = len(nverts)
    p = rng.integers(0, P, n)
    c = verts[p, rng.integers(0, 3, n)]
    o = rng.uniform(0.05, 0.95, (n, 3)) * np.asarray(size)
    d = rng.normal(size=(n, 3))
    k = rng.integers(0, 3, n)
    two = rng.random(n) < 0.3
    k2 = (k + 1 + rng.integers(0, 2, n)) % 3
    rows = np.arange(n)
    o[rows, k] = c[rows, k]; d[rows, k] = 0.0
    o[rows[two], k2[two]] = c[rows[two], k2[two]]; d[rows[two], k2[two]] = 0.0
    on = rng.random(n) < 0.15
    o[on] = c[on]
    d[np.all(d == 0, axis=1)] = (0.0, 0.0, 1.0)
    d[::7] *= -1.0
    return np.ascontiguousarray(np.concatenate([o, d], 1))


def test_tight_boxes_change_nothing_on_ties_faces_and_planes():
    v, nv, size = tie_scene()
    T, To = H.Topology(v, nv), po.Topology(v, nv)
    rays = np.concatenate([tie_rays(v, nv, size, n=5000), face_rays(v, nv, size)])
    for depth, maxp in ((4, 8), (7, 2), (0, 4)):
        oc, oo = H.Octree([T], depth, maxp), po.Octree([To], depth, maxp)
        ref = both_ways(oc, oo, rays, f"ties {depth}/{maxp}")
        e1 = ref["poly_id"].astype(np.int32)                    # leave the polygon just hit: a coincident twin answers
        both_ways(oc, oo, rays, f"ties {depth}/{maxp} excl", poly_origin1=e1)
        assert (ref["hit"] != 0).sum() > 4000


def test_tight_boxes_change_nothing_in_a_soup_with_quads_and_outside_origins():
    v, nv, size = soup(n_tri=700, n_quad=300, seed=21)
    T, To = H.Topology(v, nv), po.Topology(v, nv)
    rays = np.concatenate([soup_rays(30_000, size, seed=4), face_rays(v, nv, size, n=6000, seed=9)])
    rng = np.random.default_rng(3)
    e1 = rng.integers(-1, len(nv), len(rays)).astype(np.int32)
    e2 = rng.integers(-1, len(nv), len(rays)).astype(np.int32)
    for depth, maxp in ((5, 6), (3, 40), (8, 1)):
        oc, oo = H.Octree([T], depth, maxp), po.Octree([To], depth, maxp)
        both_ways(oc, oo, rays, f"soup {depth}/{maxp}")
        both_ways(oc, oo, rays, f"soup {depth}/{maxp} excl", poly_origin1=e1, poly_origin2=e2)


def test_far_origins_and_degenerate_directions():
    """Origins 10 ... 1e9 extents away (the guard switches the boxes off per ray beyond 1 024 extents), aimed at the scene; directions with
    components below 1e-16 (1/d becomes +1e16 whatever the sign: such rays are not `tame` and take the visit of every node), zero,
    huge and tiny magnitudes."""
    v, nv, size = soup(n_tri=500, n_quad=100, seed=5)
    T, To = H.Topology(v, nv), po.Topology(v, nv)
    rng = np.random.default_rng(8)
    n = 12_000
    tgt = rng.uniform(0.1, 0.9, (n, 3)) * np.asarray(size)
    u = rng.normal(size=(n, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    dist = 6.0 * 10.0 ** rng.integers(1, 10, n).astype(np.float64)
    o = tgt - u * dist[:, None]
    d = u * 2.0 ** rng.integers(-40, 40, n)[:, None].astype(np.float64)
    near = soup_rays(4000, size, seed=12)
    near[::5, 3] = 1e-17; near[1::5, 4] = -1e-17; near[2::5, 5] = 0.0; near[3::25, 3:] *= 1e-200; near[4::25, 3:] *= 1e200
    rays = np.ascontiguousarray(np.concatenate([np.concatenate([o, d], 1), near]))
    oc, oo = H.Octree([T], 5, 6), po.Octree([To], 5, 6)
    ref = both_ways(oc, oo, rays, "far origins")
    assert (ref["hit"] != 0).sum() > 3000


def test_two_topologies_and_a_deep_tree():
    v0, n0, size = soup()
    v1, n1, _ = soup(n_tri=300, n_quad=80, seed=9, size=(5.0, 4.5, 3.5))
    Ts, To = [H.Topology(v0, n0), H.Topology(v1, n1)], [po.Topology(v0, n0), po.Topology(v1, n1)]
    rays = soup_rays(20_000, size)
    oc, oo = H.Octree(Ts, 5, 6), po.Octree(To, 5, 6)
    for top in (0, 1):                      # the lists hold the LAST topology's ids; each topology's polygons have their own boxes
        ref, _ = oo.shoot(rays, top_index=top)
        for name, k in KERNELS.items():
            oc.set_option("octree_kernel", k)
            for tight in (1, 0):
                oc.set_option("octree_tight", tight)
                assert_events_equal(oc.Shoot_batch(rays, top_index=top)[0], ref, what=f"two topologies top {top} {name} tight={tight}")
    dv, dnv, dr = deep_scene(17)
    both_ways(H.Octree([H.Topology(dv, dnv)], 17, 1), po.Octree([po.Topology(dv, dnv)], 17, 1), dr, "deep 17")


def test_the_bench_workload_with_and_without_the_boxes():
    m = H.scenes.hall()
    T, To = H.Topology(m.verts, m.nverts), po.Topology(m.verts, m.nverts)
    rays = H.scenes.burst_rays(400_000, m.size)
    oc, oo = H.Octree([T], 8, 16), po.Octree([To], 8, 16)
    ref, rc = oo.shoot(rays, nthreads=32)
    assert oc.kernel_name(len(rays)) == "hare_octree_dense"
    for tight in (1, 0):
        oc.set_option("octree_tight", tight)
        ev, c = oc.Shoot_batch(rays)
        assert_events_equal(ev, ref, what=f"hall tight={tight}")
        assert c["hits"] == rc["hits"]


def test_kdtree_with_and_without_the_boxes():
    """KDTree.Shoot visits every leaf (KDTree.cs:210-216, F4); with the boxes `hare_kdtree_shoot` drops the subtrees the ray cannot hit.
    Same events either way and from the oracle: ties and faces, a soup with quadrilaterals and exclusions, far origins, two topologies,
    and a scene the brute-force query could not finish on the GPU in reasonable time without them (the hall, 100k polygons)."""
    def check(kd, ko, rays, what, top=0, **kw):
        okw = {("excl1" if k == "poly_origin1" else "excl2"): v for k, v in kw.items()}
        ref, rc = ko.shoot(rays, top_index=top, **okw)
        for tight in (1, 0):
            kd.set_option("octree_tight", tight)
            ev, c = kd.Shoot_batch(rays, top_index=top, **kw)
            assert_events_equal(ev, ref, what=f"kd {what} tight={tight}")
            assert c["hits"] == rc["hits"]
        kd.set_option("octree_tight", 1)
        return ref

    v, nv, size = tie_scene()
    T, To = H.Topology(v, nv), po.Topology(v, nv)
    rays = np.concatenate([tie_rays(v, nv, size, n=3000), face_rays(v, nv, size, n=2000)])
    kd, ko = H.KDTree([T], 7, 6), po.KDTree([To], 7, 6)
    ref = check(kd, ko, rays, "ties")
    check(kd, ko, rays, "ties excl", poly_origin1=ref["poly_id"].astype(np.int32))

    v, nv, size = soup(n_tri=700, n_quad=300, seed=21)
    T, To = H.Topology(v, nv), po.Topology(v, nv)
    rng = np.random.default_rng(8)
    n = 3000
    tgt = rng.uniform(0.1, 0.9, (n, 3)) * np.asarray(size)
    u = rng.normal(size=(n, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    far = np.concatenate([tgt - u * (6.0 * 10.0 ** rng.integers(1, 10, n).astype(np.float64))[:, None], u], 1)
    near = soup_rays(5000, size, seed=12)
    near[::5, 3] = 1e-17; near[1::5, 4] = 0.0; near[2::5, 5] = -0.0; near[3::25, 3:] *= 1e-200; near[4::25, 3:] *= 1e200
    rays = np.ascontiguousarray(np.concatenate([near, far, face_rays(v, nv, size, n=2000, seed=5)]))
    e1 = rng.integers(-1, len(nv), len(rays)).astype(np.int32)
    for depth, maxp in ((6, 8), (12, 1), (0, 4)):
        kd, ko = H.KDTree([T], depth, maxp), po.KDTree([To], depth, maxp)
        check(kd, ko, rays, f"soup {depth}/{maxp}")
        check(kd, ko, rays, f"soup {depth}/{maxp} excl", poly_origin1=e1)

    v0, n0, size = soup()
    v1, n1, _ = soup(n_tri=300, n_quad=80, seed=9, size=(5.0, 4.5, 3.5))
    kd, ko = H.KDTree([H.Topology(v0, n0), H.Topology(v1, n1)], 6, 8), po.KDTree([po.Topology(v0, n0), po.Topology(v1, n1)], 6, 8)
    for top in (0, 1):
        check(kd, ko, soup_rays(4000, size), f"two topologies top {top}", top=top)

    m = H.scenes.hall()
    rays = H.scenes.burst_rays(1 << 20, m.size)[::512]              # 2 048 rays: the oracle's query is 100k polygons per ray
    kd, ko = H.KDTree([H.Topology(m.verts, m.nverts)], 16, 8), po.KDTree([po.Topology(m.verts, m.nverts)], 16, 8)
    ref, rc = ko.shoot(rays, nthreads=32)
    ev, c = kd.Shoot_batch(rays)
    assert_events_equal(ev, ref, what="kd hall")
    assert c["hits"] == rc["hits"] == len(rays)
