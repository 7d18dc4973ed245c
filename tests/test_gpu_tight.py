"""The subtree TIGHT BOXES of the octree kernels K2d / K2p / K2g (device_scene.cpp: make_tight_boxes; scene option `octree_tight`): a popped node whose
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
        for kern, kname in ((2, "hare_kdtree_dense"), (1, "hare_kdtree_shoot")):      # K3d (round 5) and the one-ray-per-lane kernel
            kd.set_option("kdtree_kernel", kern)
            assert kd.kernel_name(len(rays), top) == kname
            for tight in (1, 0):
                kd.set_option("octree_tight", tight)
                ev, c = kd.Shoot_batch(rays, top_index=top, **kw)
                assert_events_equal(ev, ref, what=f"kd {what} {kname} tight={tight}")
                assert c["hits"] == rc["hits"]
        kd.set_option("octree_tight", 1)
        kd.set_option("kdtree_kernel", 0)
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
    for kern in (2, 1):
        kd.set_option("kdtree_kernel", kern)
        ev, c = kd.Shoot_batch(rays)
        assert_events_equal(ev, ref, what=f"kd hall kernel {kern}")
        assert c["hits"] == rc["hits"] == len(rays)


# ---------------------------------------------------------------------------------------------------------------- Voxel_Grid
def voxel_both_ways(g, o, rays, what, **kw):
    """K1q with the voxels' tight boxes on and off against the oracle (scene option voxel_tight)."""
    okw = {("excl1" if k == "poly_origin1" else "excl2"): v for k, v in kw.items()}
    ref, rc = o.shoot(rays, **okw)
    for tight in (1, 0):
        g.set_option("voxel_tight", tight)
        ev, c = g.Shoot_batch(rays, **kw)
        assert_events_equal(ev, ref, what=f"voxel {what} tight={tight}")
        assert c["hits"] == rc["hits"]
    g.set_option("voxel_tight", 1)
    return ref


def test_voxel_tight_boxes_change_nothing():
    """A ray without a hit that meets an occupied voxel whose polygons it cannot hit walks on without scanning the list (voxel_pool.hip).
    Ties / faces / planes, a soup with quadrilaterals, exclusions, origins outside the grid (AABB.Intersect moves them: the box test uses
    the moved origin) with and without the origin write-back, far origins and degenerate directions, one voxel ... a coarse bitmap."""
    v, nv, size = tie_scene()
    T, To = H.Topology(v, nv), po.Topology(v, nv)
    rays = np.concatenate([tie_rays(v, nv, size, n=5000), face_rays(v, nv, size)])
    for D in (1, 8, 21, 64):
        g, o = H.Voxel_Grid([T], D), po.VoxelGrid([To], domain=D)
        assert g.kernel_name(len(rays)).startswith("hare_voxel_pool")
        ref = voxel_both_ways(g, o, rays, f"ties D={D}")
        voxel_both_ways(g, o, rays, f"ties D={D} excl", poly_origin1=ref["poly_id"].astype(np.int32))

    v, nv, size = soup(n_tri=700, n_quad=300, seed=21)
    T, To = H.Topology(v, nv), po.Topology(v, nv)
    rng = np.random.default_rng(8)
    n = 6000
    tgt = rng.uniform(0.1, 0.9, (n, 3)) * np.asarray(size)
    u = rng.normal(size=(n, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    far = np.concatenate([tgt - u * (6.0 * 10.0 ** rng.integers(1, 10, n).astype(np.float64))[:, None], u * 2.0 ** rng.integers(-30, 30, n)[:, None]], 1)
    near = soup_rays(30_000, size, seed=4)
    near[::5, 3] = 1e-17; near[1::5, 4] = 0.0; near[2::5, 5] = -0.0; near[3::25, 3:] *= 1e-200; near[4::25, 3:] *= 1e200
    rays = np.ascontiguousarray(np.concatenate([near, far, face_rays(v, nv, size, n=6000, seed=9)]))
    e1 = rng.integers(-1, len(nv), len(rays)).astype(np.int32)
    e2 = rng.integers(-1, len(nv), len(rays)).astype(np.int32)
    for D in (5, 12, 31, 128):                 # 128: one occupancy bit per block of voxels (the `_g` kernels)
        g, o = H.Voxel_Grid([T], D), po.VoxelGrid([To], domain=D)
        voxel_both_ways(g, o, rays, f"soup D={D}")
        voxel_both_ways(g, o, rays, f"soup D={D} excl", poly_origin1=e1, poly_origin2=e2)
        refm, _, moved = o.shoot(rays, mutate=True)
        for tight in (1, 0):
            g.set_option("voxel_tight", tight)
            r = rays.copy()
            ev, _ = g.Shoot_batch(r, writeback_origin=True)
            assert_events_equal(ev, refm, what=f"soup D={D} write-back tight={tight}")
            assert np.array_equal(r.view(np.int64), moved.view(np.int64))
        g.set_option("voxel_tight", 1)


def test_voxel_tight_boxes_in_the_bounce_loop_and_over_two_topologies():
    """Reflected rays start ON a polygon -- inside its voxel's box -- and skim their wall: where the boxes reject most.  Cast by cast
    against the oracle's loop, boxes on and off, a launch per cast and the one-launch loop; then a grid over two topologies."""
    from tests.helpers import oracle_bounce_loop
    m = H.scenes.hall()
    T, To = H.Topology(m.verts, m.nverts), po.Topology(m.verts, m.nverts)
    rays = H.scenes.burst_rays(120_000, m.size)
    g, o = H.Voxel_Grid([T], 64), po.VoxelGrid([To], domain=64)
    ref, rc = oracle_bounce_loop(po, To, o, rays, 6)
    for tight in (1, 0):
        g.set_option("voxel_tight", tight)
        for fused in (0, 1):
            g.set_option("bounce_fused", fused)
            ev, c, pcs = g.Bounce_batch(rays, 6, per_cast=True, all_casts=True)         # every cast's events (a launch per cast, packing)
            for b in range(6):
                assert_events_equal(ev[b], ref[b], what=f"bounce cast {b} tight={tight} fused={fused}")
            assert [(p["rays"], p["hits"]) for p in pcs] == [(p["rays"], p["hits"]) for p in rc]
            last, c, pcs = g.Bounce_batch(rays, 6, per_cast=True)                       # the last cast only (one launch when fused)
            assert last.tobytes() == ref[5].tobytes(), (tight, fused)
            assert [(p["rays"], p["hits"]) for p in pcs] == [(p["rays"], p["hits"]) for p in rc]
    g.set_option("bounce_fused", 0); g.set_option("voxel_tight", 1)

    v0, n0, size = soup()
    v1, n1, _ = soup(n_tri=300, n_quad=80, seed=9, size=(5.0, 4.5, 3.5))
    g = H.Voxel_Grid([H.Topology(v0, n0), H.Topology(v1, n1)], 10)
    o = po.VoxelGrid([po.Topology(v0, n0), po.Topology(v1, n1)], domain=10)
    rays = soup_rays(20_000, size)
    for top in (0, 1):
        ref, _ = o.shoot(rays, top_index=top)
        for tight in (1, 0):
            g.set_option("voxel_tight", tight)
            assert_events_equal(g.Shoot_batch(rays, top_index=top)[0], ref, what=f"voxel two topologies top {top} tight={tight}")


def test_rays_that_graze_polygon_boxes_by_less_than_the_margin():
    """Where the boxes have the least room: targets ON polygon edges and corners (= on the faces and corners of the polygons' boxes) moved
    by 0, 1e-15 ... 1e-3 in a random direction, from origins up to a thousand extents away (the range the margin of 2^-20 of the extent is
    sized for; the rounding of the exact test grows with that distance) and with directions of every magnitude.  Whether such a ray hits
    is the exact test's business; the boxes must never decide it.  Octree (all kernels), kd-tree and Voxel_Grid against the oracle."""
    v, nv, size = tie_scene()
    T, To = H.Topology(v, nv), po.Topology(v, nv)
    rng = np.random.default_rng(77)
    n = 24_000
    p = rng.integers(0, len(nv), n)
    k0 = rng.integers(0, 3, n); k1 = (k0 + 1) % 3
    A, B = v[p, k0], v[p, k1]
    w = rng.choice([0.0, 0.25, 0.5, 1.0], n)[:, None]
    eps = rng.choice([0.0, 1e-15, 1e-13, 1e-11, 1e-9, 1e-7, 1e-5, 1e-3], n)[:, None]
    u = rng.normal(size=(n, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    tgt = A + (B - A) * w + u * eps
    dist = np.asarray(size).max() * rng.choice([0.3, 1.0, 10.0, 100.0, 1000.0], n)[:, None]
    e = rng.normal(size=(n, 3)); e /= np.linalg.norm(e, axis=1, keepdims=True)
    o = tgt - e * dist
    d = (tgt - o) * 2.0 ** rng.integers(-20, 20, n)[:, None].astype(np.float64)
    rays = np.ascontiguousarray(np.concatenate([o, d], 1))
    oc, oo = H.Octree([T], 6, 4), po.Octree([To], 6, 4)
    ref = both_ways(oc, oo, rays, "grazing octree")
    assert 0.3 * n < (ref["hit"] != 0).sum() < n
    kd, ko = H.KDTree([T], 8, 4), po.KDTree([To], 8, 4)
    kref, krc = ko.shoot(rays[:8000])
    for tight in (1, 0):
        kd.set_option("octree_tight", tight)
        assert_events_equal(kd.Shoot_batch(rays[:8000])[0], kref, what=f"grazing kd tight={tight}")
    for D in (8, 32):
        voxel_both_ways(H.Voxel_Grid([T], D), po.VoxelGrid([To], domain=D), rays, f"grazing D={D}")


def test_a_scene_far_from_the_origin_of_its_coordinates():
    """The boxes' margin is 2^-20 of the scene's extent OR of its largest coordinate, whichever is larger: a model placed at 1e6 or 1e9
    (coordinates on the 2^-8 lattice stay exact there) has the rounding of its coordinates in every test, exact or not.  Same events with
    the boxes on, off and from the oracle, on every partition."""
    v, nv, size = soup(n_tri=600, n_quad=150, seed=31)
    rays0 = np.concatenate([soup_rays(12_000, size, seed=6), face_rays(v, nv, size, n=3000, seed=7)])
    for shift in ((2.0 ** 20, -2.0 ** 21, 2.0 ** 19), (2.0 ** 30, 2.0 ** 30, -2.0 ** 29)):
        sh = np.asarray(shift)
        vs = v.copy()
        for p in range(len(nv)):
            vs[p, :nv[p]] += sh
        rays = rays0.copy(); rays[:, :3] += sh
        T, To = H.Topology(vs, nv), po.Topology(vs, nv)
        for D in (6, 20):
            voxel_both_ways(H.Voxel_Grid([T], D), po.VoxelGrid([To], domain=D), rays, f"shifted {shift[0]:.0e} D={D}")
        both_ways(H.Octree([T], 5, 6), po.Octree([To], 5, 6), rays, f"shifted {shift[0]:.0e} octree")
        kd, ko = H.KDTree([T], 7, 8), po.KDTree([To], 7, 8)
        kref, _ = ko.shoot(rays[:6000])
        for tight in (1, 0):
            kd.set_option("octree_tight", tight)
            assert_events_equal(kd.Shoot_batch(rays[:6000])[0], kref, what=f"shifted {shift[0]:.0e} kd tight={tight}")


def test_a_grid_without_its_tight_boxes_is_built_and_traced_all_the_same(monkeypatch):
    """ADVICE (round 4): the boxes cost 32 B per voxel and topology and are an acceleration, not a precondition.  They are built only while
    `voxel_tight` is on and the pool kernel serves the grid; over the budget `voxel_tight_max_mb`, or when their allocation fails (the
    hook `dev_fail_cellbox_alloc` / HARE_FAIL_CELLBOX_ALLOC), the build SUCCEEDS, the scene holds no boxes and every X_Event is the same."""
    v, nv, size = soup(n_tri=600, n_quad=100, seed=5)
    T, To = H.Topology(v, nv), po.Topology(v, nv)
    rays = soup_rays(20_000, size, seed=6)
    D = 24
    ref, _ = po.VoxelGrid([To], domain=D).shoot(rays)
    g = H.Voxel_Grid([T], D)
    full = D * D * D * 32
    assert g.get_option("voxel_tight_bytes") == full                     # built with the grid
    assert_events_equal(g.Shoot_batch(rays)[0], ref, what="boxes built")
    g.set_option("voxel_tight_max_mb", 0)
    g.set_option("dev_fail_cellbox_alloc", 1)                            # "out of memory": the call succeeds, no boxes
    assert g.get_option("voxel_tight_bytes") == 0
    assert_events_equal(g.Shoot_batch(rays)[0], ref, what="allocation failed")
    g.set_option("dev_fail_cellbox_alloc", 0)                            # memory is back: they are rebuilt
    assert g.get_option("voxel_tight_bytes") == full
    g2 = H.Voxel_Grid([T], 128)                                          # 64 MiB of boxes against a budget of 1 MiB
    g2.set_option("voxel_tight_max_mb", 1)
    assert g2.get_option("voxel_tight_bytes") == 0
    ref128, _ = po.VoxelGrid([To], domain=128).shoot(rays)
    assert_events_equal(g2.Shoot_batch(rays)[0], ref128, what="over budget")
    g2.set_option("voxel_tight_max_mb", 0)
    assert g2.get_option("voxel_tight_bytes") == 128 ** 3 * 32
    assert_events_equal(g2.Shoot_batch(rays)[0], ref128, what="budget lifted")
    # a grid BUILT while the allocation fails (the environment is read when the scene is created; developer variables need HARE_DEV)
    monkeypatch.setenv("HARE_DEV", "1")
    monkeypatch.setenv("HARE_FAIL_CELLBOX_ALLOC", "1")
    g3 = H.Voxel_Grid([T], D)
    assert g3.get_option("voxel_tight_bytes") == 0
    assert_events_equal(g3.Shoot_batch(rays)[0], ref, what="built without boxes")
    monkeypatch.delenv("HARE_FAIL_CELLBOX_ALLOC")
    # switched off before the build: nothing is allocated; switched on later: built then
    monkeypatch.setenv("HARE_VOXEL_TIGHT", "0")
    g4 = H.Voxel_Grid([T], D)
    assert g4.get_option("voxel_tight") == 0 and g4.get_option("voxel_tight_bytes") == 0
    assert_events_equal(g4.Shoot_batch(rays)[0], ref, what="option off")
    g4.set_option("voxel_tight", 1)
    assert g4.get_option("voxel_tight_bytes") == full
    assert_events_equal(g4.Shoot_batch(rays)[0], ref, what="option on after the build")
