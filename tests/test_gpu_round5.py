"""Round 5 on the GPU: K3d `hare_kdtree_dense` (kdtree_dense.hip) -- KDTree.Shoot (KDTree.cs:198-361) as a production kernel: persistent
waves, one-line node records with both children's tight boxes, leaves pre-culled densely, exact tests deferred and made in the reference's
visiting order.  Every case compares all eight X_Event fields with the oracle, with K3d and with the one-ray-per-lane kernel it replaces
(`kdtree_kernel` 2 / 1): ties (where the visiting ORDER decides the polygon), trees from a single leaf to 26 levels, one polygon per leaf,
exclusions, degenerate and far rays, two topologies, batches of 1 ray ... 300k, the bounce loop cast by cast, and the counting build."""
import numpy as np
import pytest
import torch

import hare_amd as H
from hare_amd import capi
from oracle import pyoracle as po
from tests.helpers import assert_events_equal, oracle_bounce_loop, soup, soup_rays
from tests.test_gpu_ties import tie_rays, tie_scene

pytestmark = pytest.mark.gpu
KERNELS = ((2, "hare_kdtree_dense"), (1, "hare_kdtree_shoot"))


def both(kd, ko, rays, what, top=0, **kw):
    okw = {("excl1" if k == "poly_origin1" else "excl2"): v for k, v in kw.items()}
    ref, rc = ko.shoot(rays, top_index=top, **okw)
    for kern, name in KERNELS:
        kd.set_option("kdtree_kernel", kern)
        assert kd.kernel_name(len(rays), top) == name
        ev, c = kd.Shoot_batch(rays, top_index=top, **kw)
        assert_events_equal(ev, ref, what=f"{what} {name}")
        assert c["hits"] == rc["hits"] and c["rays"] == len(rays)
    kd.set_option("kdtree_kernel", 0)
    return ref


def test_k3d_is_the_default_and_ties_go_to_the_polygon_the_reference_meets_first():
    v, nv, size = tie_scene()
    T, To = H.Topology(v, nv), po.Topology(v, nv)
    rays = tie_rays(v, nv, size, n=6000)
    for depth, maxp in ((7, 6), (10, 1), (3, 40)):
        kd, ko = H.KDTree([T], depth, maxp), po.KDTree([To], depth, maxp)
        assert kd.kernel_name(len(rays)) == "hare_kdtree_dense"                 # the library's rule
        ref = both(kd, ko, rays, f"ties {depth}/{maxp}")
        both(kd, ko, rays, f"ties {depth}/{maxp} excl", poly_origin1=ref["poly_id"].astype(np.int32))
        e2 = np.roll(ref["poly_id"], 1).astype(np.int32)
        both(kd, ko, rays, f"ties {depth}/{maxp} excl x2", poly_origin1=ref["poly_id"].astype(np.int32), poly_origin2=e2)


def test_k3d_tree_shapes_from_one_leaf_to_20_levels():
    """maxDepth 0 (the root is the only leaf) ... 20 with one polygon per leaf (lists double where polygons straddle a split: 1.8M nodes at 20
    levels; the stack is depth + 2 entries per lane in LDS, 49 KB per workgroup there)."""
    v, nv, size = soup(n_tri=150, n_quad=50, seed=17)
    T, To = H.Topology(v, nv), po.Topology(v, nv)
    rays = soup_rays(9000, size, seed=3)
    rays[::7, 3] = 0.0; rays[1::7, 4] = -0.0; rays[2::7, 5] = 1e-17; rays[3::49, 3:] *= 1e-200; rays[4::49, 3:] *= 1e200; rays[5::49, 0] = np.nan
    for depth, maxp in ((0, 4), (1, 1), (5, 8), (14, 2), (20, 1)):
        try:
            ko = po.KDTree([To], depth, maxp)
        except MemoryError:
            continue                                                            # the oracle's budget: lists that double per level
        kd = H.KDTree([T], depth, maxp)
        assert kd.info().n_nodes == ko.n_nodes
        # the oracle visits EVERY leaf for every ray (F4): on a big tree it gets fewer rays
        m = len(rays) if ko.n_nodes < 3000 else max(300, min(len(rays), 40_000_000 // ko.n_nodes))
        both(kd, ko, rays[:m], f"shape {depth}/{maxp} ({ko.n_nodes} nodes, {m} rays)")


def test_k3d_far_origins_two_topologies_and_boxes_off():
    v, nv, size = soup(n_tri=700, n_quad=300, seed=21)
    T, To = H.Topology(v, nv), po.Topology(v, nv)
    rng = np.random.default_rng(8)
    n = 4000
    tgt = rng.uniform(0.1, 0.9, (n, 3)) * np.asarray(size)
    u = rng.normal(size=(n, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    far = np.concatenate([tgt - u * (6.0 * 10.0 ** rng.integers(1, 10, n).astype(np.float64))[:, None], u * 2.0 ** rng.integers(-30, 30, n)[:, None]], 1)
    rays = np.ascontiguousarray(np.concatenate([soup_rays(6000, size, seed=12), far]))
    kd, ko = H.KDTree([T], 9, 4), po.KDTree([To], 9, 4)
    both(kd, ko, rays, "far origins")
    kd.set_option("octree_tight", 0)                      # no box test at all: every node is visited, as the reference does
    both(kd, ko, rays, "far origins, boxes off")
    kd.set_option("octree_tight", 1)
    v1, n1, _ = soup(n_tri=300, n_quad=80, seed=9, size=(5.0, 4.5, 3.5))
    v0, n0, size0 = soup()
    kd2, ko2 = H.KDTree([H.Topology(v0, n0), H.Topology(v1, n1)], 6, 8), po.KDTree([po.Topology(v0, n0), po.Topology(v1, n1)], 6, 8)
    for top in (0, 1):
        both(kd2, ko2, soup_rays(5000, size0), f"two topologies top {top}", top=top)


def test_k3d_batches_of_every_size_and_many_launches_in_flight():
    m = H.scenes.shoebox()
    T, To = H.Topology(m.verts, m.nverts), po.Topology(m.verts, m.nverts)
    kd, ko = H.KDTree([T], 12, 16), po.KDTree([To], 12, 16)
    rays = H.scenes.random_rays(300_000, m.size)
    ref, _ = ko.shoot(rays, nthreads=16)
    for n in (1, 2, 63, 64, 65, 255, 4097, 65536, 196_609, 300_000):
        ev, c = kd.Shoot_batch(rays[:n])
        assert_events_equal(ev, ref[:n], what=f"n={n}")
        assert c["rays"] == n
    # 200 launches over four streams without a host synchronisation (the launch-slot ring comes round three times)
    d_rays = torch.from_numpy(rays[:20_000].copy()).cuda()
    outs = [torch.empty(20_000 * 56, dtype=torch.uint8, device="cuda") for _ in range(4)]
    streams = [torch.cuda.Stream() for _ in range(4)]
    torch.cuda.synchronize()
    for k in range(200):
        kd.shoot_device(20_000, d_rays.data_ptr(), outs[k % 4].data_ptr(), stream=streams[k % 4].cuda_stream)
    torch.cuda.synchronize()
    for o in outs:
        got = np.frombuffer(o.cpu().numpy().tobytes(), dtype=capi.XEVENT_DTYPE)
        assert_events_equal(got, ref[:20_000], what="launches in flight")


def test_k3d_in_the_bounce_loop_and_its_counting_build():
    """hare_bounce_batch over a KDTree runs a launch per cast with the retired rays skipped (flag HARE_SHOOT_RETIRED_RAYS): every cast against
    the oracle's loop, in an open soup where rays die.  Then HARE_SHOOT_COUNT_OWN: the counting build returns the same events and counters
    that make sense (every ray fetches at least the root, pre-culls >= exact tests)."""
    v, nv, size = soup(n_tri=500, n_quad=150, seed=33)
    T, To = H.Topology(v, nv), po.Topology(v, nv)
    kd, ko = H.KDTree([T], 8, 6), po.KDTree([To], 8, 6)
    rays = soup_rays(30_000, size, seed=2)
    ref_all, _ = oracle_bounce_loop(po, To, ko, rays, 5)
    for kern, name in KERNELS:
        kd.set_option("kdtree_kernel", kern)
        ev_all, ctr = kd.Bounce_batch(rays, 5, all_casts=True)
        for b in range(5):
            assert_events_equal(ev_all[b], ref_all[b], what=f"bounce cast {b} {name}")
    kd.set_option("kdtree_kernel", 0)
    n = len(rays)
    d_rays = torch.from_numpy(rays).cuda()
    d_out = torch.empty(n * 56, dtype=torch.uint8, device="cuda")
    d_ctr = torch.zeros(8, dtype=torch.int64, device="cuda")
    assert kd.kernel_name(n, flags=capi.SHOOT_COUNT_OWN) == "hare_kdtree_dense_own"
    kd.shoot_device(n, d_rays.data_ptr(), d_out.data_ptr(), d_counters=d_ctr.data_ptr(), flags=capi.SHOOT_COUNT_OWN)
    torch.cuda.synchronize()
    got = np.frombuffer(d_out.cpu().numpy().tobytes(), dtype=capi.XEVENT_DTYPE)
    assert_events_equal(got, ref_all[0], what="counting build")
    c = [int(x) for x in d_ctr.cpu()]
    assert c[0] == n and c[1] == int(ref_all[0]["hit"].sum())
    assert c[2] > 0 and c[3] == c[5] and c[5] >= c[4] > 0                # nodes fetched, entries == pre-culls >= exact tests
    _, ref_ctr = ko.shoot(rays)
    assert c[2] < ref_ctr["cells"] and c[4] < ref_ctr["tests"]           # ... and far fewer than the reference's visit of every leaf
    kd.set_option("kdtree_kernel", 1)
    with pytest.raises(H.HareError) as e:                                 # the one-ray-per-lane kernel has no counting build: the call says so
        kd.shoot_device(n, d_rays.data_ptr(), d_out.data_ptr(), d_counters=d_ctr.data_ptr(), flags=capi.SHOOT_COUNT_OWN)
    assert e.value.code == capi.HARE_E_UNSUPPORTED


def test_the_cost_order_of_the_pool_kernel_changes_no_event():
    """order_kernels.hip: K1q takes the rays of a batch, window by window of 4 096, in the order of their estimated walk length
    (`voxel_order`: 1 = batches of primary rays from 1 572 864 rays, 0 never, 2 every batch).  Rays, events and exclusions stay in the
    caller's order and every X_Event is the oracle's: batch sizes around the windows, origins outside the grid with the origin write-back,
    exclusion arrays (forced: the rule leaves such batches alone), NaN / zero / huge direction components, a coarse bitmap."""
    v, nv, size = soup(n_tri=900, n_quad=200, seed=41)
    T, To = H.Topology(v, nv), po.Topology(v, nv)
    rays = soup_rays(300_000, size, seed=6)
    rays[::11, 3] = 0.0; rays[1::11, 4] = -0.0; rays[2::13, 5] = 1e-300; rays[3::97, 3:] *= 1e200; rays[5::101, 0] = np.nan; rays[7::103, 3] = np.inf
    for D in (24, 96):
        g, o = H.Voxel_Grid([T], D), po.VoxelGrid([To], domain=D)
        ref, rc = o.shoot(rays, nthreads=16)
        refm, _, moved = o.shoot(rays, nthreads=16, mutate=True)
        for n in (1, 4095, 4096, 4097, 262_143, 262_144, 300_000):
            for order in (2, 1, 0):
                g.set_option("voxel_order", order)
                ev, c = g.Shoot_batch(rays[:n])
                assert_events_equal(ev, ref[:n], what=f"D={D} n={n} order={order}")
                assert c["rays"] == n
        e1 = ref["poly_id"].astype(np.int32)
        e2 = np.roll(e1, 3)
        ref2, _ = o.shoot(rays, excl1=e1, excl2=e2, nthreads=16)
        for order in (2, 0):
            g.set_option("voxel_order", order)
            assert_events_equal(g.Shoot_batch(rays, poly_origin1=e1, poly_origin2=e2)[0], ref2, what=f"D={D} excl order={order}")
            r = rays.copy()
            ev, _ = g.Shoot_batch(r, writeback_origin=True)
            assert_events_equal(ev, refm, what=f"D={D} write-back order={order}")
            nan = np.isnan(moved)                                   # a NaN / infinite ray may come back with a NaN origin: the payload bits are the FPU's
            assert np.array_equal(np.isnan(r), nan) and np.array_equal(r.view(np.int64)[~nan], moved.view(np.int64)[~nan])
        g.set_option("voxel_order", 1)


def test_the_quadrilateral_pre_cull_never_rejects_a_hit(monkeypatch):
    """Round 5: a quadrilateral is pre-culled when cull_fp32 is certain to miss BOTH triangles Quadrilateral.Intersect tries
    (Hare_Geometry_Polygons.cs:784-823: (0,1,2) then (2,3,0)); until now it went to the exact test unfiltered.  The audit kernel runs every
    ray against every polygon through the production load / decode / test and counts (culled AND the exact test accepts): must be zero --
    on general convex quads (not only parallelograms), slightly non-planar ones, slivers, scales from 1e-3 to 300 with offsets to 1e5,
    mixed with triangles; rays aimed at corners, edge points, the shared diagonal, and the interior."""
    import ctypes as C
    monkeypatch.setenv("HARE_DEV", "1")
    rng = np.random.default_rng(11)
    cases = []
    hq = H.scenes.hall_quads(edge=1.0)
    cases.append((hq.verts, hq.nverts, H.scenes.burst_rays(1500, hq.size)))
    for scale, shift in ((1.0, 0.0), (1e-3, 0.0), (1.0, 5000.0), (300.0, -1e5)):
        P = 500
        c = rng.uniform(0, 10, (P, 1, 3))
        a = rng.normal(size=(P, 3)); b = rng.normal(size=(P, 3))
        uv = np.array([[0.0, 0.0], [1.0, 0.0], [1.0, 1.0], [0.0, 1.0]])[None] + rng.uniform(-0.3, 0.3, (P, 4, 2))     # convex-ish corners
        quad = c + uv[..., :1] * a[:, None, :] + uv[..., 1:] * b[:, None, :]
        quad[::5] += rng.normal(size=(P, 4, 3))[::5] * 0.02                       # a fifth slightly non-planar
        quad[3::9, 3] = quad[3::9, 0] + (quad[3::9, 2] - quad[3::9, 0]) * 1.00001   # slivers: second triangle nearly degenerate
        verts = quad * scale + shift
        nverts = np.full(P, 4, np.int32)
        nverts[::4] = 3                                                           # a quarter are triangles (corner 3 ignored)
        o = rng.uniform(0, 10, (3000, 3)) * scale + shift
        pick = rng.integers(0, P, 3000)
        w = rng.dirichlet([0.3, 0.3, 0.3, 0.3], 3000)
        w[::5] = np.eye(4)[rng.integers(0, 4, 600)]                               # exactly at a corner
        w[1::5, 1] = 0; w[1::5, 3] = 0; w[1::5] /= w[1::5].sum(1, keepdims=True)  # on the diagonal v0 - v2 both triangles share
        aim = np.einsum("nk,nkc->nc", w, verts[pick])
        d = aim - o
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        cases.append((np.ascontiguousarray(verts), nverts, np.concatenate([o, d], 1)))
    for verts, nverts, rays in cases:
        g = H.Voxel_Grid([H.Topology(verts, nverts)], 2)
        rays = np.ascontiguousarray(rays)
        out = np.zeros(len(rays), capi.XEVENT_DTYPE)
        ctr = (C.c_uint64 * 8)()
        capi.check(capi.lib.hare_shoot_batch(g._h, 0, 0, len(rays), rays.ctypes.data, None, None, 0x8000, out.ctypes.data, C.addressof(ctr)))
        viol, culled, cands = ctr[5], ctr[6], ctr[7]
        assert cands == len(rays) * g.Model[0].Polygon_Count
        assert viol == 0, f"the pre-cull rejected {viol} true hits on quadrilaterals"
        assert culled > 0.5 * cands                                               # ... and quadrilaterals ARE culled now


def test_kdtree_occlusion_flags_from_k3d_stop_early_and_agree_with_the_closest_hit():
    """hare_kdtree_occl (K3d's OCC build): occluded = KDTree.Shoot hits and that closest hit has t < t_max.  The flags-only kernel ends a
    ray at the first polygon accepted below t_max and skips subtrees the ray enters at or beyond t_max; the flag must equal the one
    derived from the oracle's closest hit for t_max at 0.3 ... 3 x every ray's own hit distance, EXACTLY at it and one ulp either side,
    infinite, NaN, negative, zero -- with exclusions, on the hall (every 512th burst ray) and an open soup, from both kernels."""
    cases = []
    m = H.scenes.hall()
    cases.append((m.verts, m.nverts, H.scenes.burst_rays(1 << 20, m.size)[::512], (16, 8)))
    v, nv, size = soup(n_tri=900, n_quad=300, seed=19)
    cases.append((v, nv, soup_rays(20_000, size, seed=7), (9, 4)))
    rng = np.random.default_rng(12)
    for verts, nverts, rays, (depth, maxp) in cases:
        kd, ko = H.KDTree([H.Topology(verts, nverts)], depth, maxp), po.KDTree([po.Topology(verts, nverts)], depth, maxp)
        n = len(rays)
        e1 = rng.integers(-1, len(nverts), n).astype(np.int32)
        for excl in (None, e1):
            ref, _ = ko.shoot(rays, excl1=excl, nthreads=16)
            t = ref["t"].copy()
            tmax = t * rng.choice([0.3, 0.999999, 1.0, 1.000001, 3.0], n)
            tmax[1::7] = t[1::7]                                         # exactly the hit distance: not occluded (strict <)
            tmax[2::7] = np.nextafter(t[2::7], np.inf)                   # one ulp beyond: occluded
            tmax[3::11] = np.inf; tmax[4::11] = np.nan; tmax[5::11] = -1.0; tmax[6::11] = 0.0
            miss = ref["hit"] == 0
            tmax[miss] = rng.uniform(0.1, 30.0, int(miss.sum()))
            want = ((ref["hit"] != 0) & (ref["t"] < tmax)).astype(np.int32)
            kw = {} if excl is None else {"poly_origin1": excl}
            for kern, name in ((0, "hare_kdtree_occl"), (1, "hare_kdtree_shoot")):
                kd.set_option("kdtree_kernel", kern)
                occ, c = kd.Occluded_batch(rays, tmax, events=False, **kw)
                bad = np.nonzero(occ != want)[0]
                assert bad.size == 0, (name, bad[:5], tmax[bad[:5]], ref[bad[:5]])
                assert c["hits"] == int(want.sum())
                occ_any, _ = kd.Occluded_batch(rays, None, events=False, **kw)
                assert np.array_equal(occ_any, (ref["hit"] != 0).astype(np.int32)), name
            kd.set_option("kdtree_kernel", 0)
            ev_occ, ev = kd.Occluded_batch(rays, tmax, events=True, **kw)      # with events: the closest-hit cast + one compare
            assert np.array_equal(ev_occ, want)
            assert_events_equal(ev, ref, what="kd occlusion with events")
