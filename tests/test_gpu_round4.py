"""Round 4, on the GPU: the bounce loop as ONE launch (hare_bounce_device, hare_voxel_bounce_*: voxel_pool.hip BOUNCE) against the
oracle's loop cast by cast, and against the launch-per-cast loop it replaces.

Reference seam: Spatial_Partition.cs:33 (Shoot with poly_origin1 = the polygon just hit), Voxel_Grid.cs:351,477; the reflection is
harness-defined (about Hare_Geometry_Polygons.cs:161-171's normal)."""
import numpy as np
import pytest

import hare_amd as H
from hare_amd import capi
from oracle import pyoracle as po
from tests.helpers import assert_events_equal, oracle_bounce_loop, soup, soup_rays

pytestmark = pytest.mark.gpu


def bounce_on_device(g, rays, bounces, excl1=None, excl2=None, all_casts=True, flags=0):
    """hare_bounce_device through torch buffers -> (events [bounces, n] or [n], per-cast counters, totals)."""
    import torch
    n = len(rays)
    d_rays = torch.from_numpy(np.ascontiguousarray(rays)).cuda()
    d_work = torch.zeros(2 * max(n, 1), dtype=torch.int32, device="cuda")
    d_last = torch.zeros(max(n, 1) * 56, dtype=torch.uint8, device="cuda")
    d_all = torch.zeros(max(n, 1) * 56 * bounces, dtype=torch.uint8, device="cuda") if all_casts else None
    d_ctr = torch.zeros(8, dtype=torch.int64, device="cuda")
    d_pc = torch.zeros(8 * bounces, dtype=torch.int64, device="cuda")
    e1 = None if excl1 is None else torch.from_numpy(np.ascontiguousarray(excl1, np.int32)).cuda()
    e2 = None if excl2 is None else torch.from_numpy(np.ascontiguousarray(excl2, np.int32)).cuda()
    g.bounce_device(n, d_rays.data_ptr(), bounces, d_work.data_ptr(), d_events_last=d_last.data_ptr(),
                    d_events_all=0 if d_all is None else d_all.data_ptr(), d_excl1=0 if e1 is None else e1.data_ptr(),
                    d_excl2=0 if e2 is None else e2.data_ptr(), d_counters=d_ctr.data_ptr(), d_counters_per_cast=d_pc.data_ptr(),
                    stream=torch.cuda.current_stream().cuda_stream, flags=flags)
    torch.cuda.synchronize()
    last = np.frombuffer(d_last.cpu().numpy().tobytes(), capi.XEVENT_DTYPE)[:n]
    allc = None if d_all is None else np.frombuffer(d_all.cpu().numpy().tobytes(), capi.XEVENT_DTYPE).reshape(bounces, max(n, 1))[:, :n]
    pc = d_pc.cpu().numpy().reshape(bounces, 8)
    tot = d_ctr.cpu().numpy()
    return allc, last, [{"rays": int(r[0]), "hits": int(r[1])} for r in pc], {"rays": int(tot[0]), "hits": int(tot[1])}


def check(g, To, o, rays, bounces, excl1=None, excl2=None, what=""):
    ref, rc = oracle_bounce_loop(po, To, o, rays, bounces, excl1=excl1, excl2=excl2)
    for fused in (1, 0):
        g.set_option("bounce_fused", fused)
        allc, last, pc, tot = bounce_on_device(g, rays, bounces, excl1, excl2)
        for b in range(bounces):
            assert_events_equal(allc[b], ref[b], what=f"{what} fused={fused} cast {b}")
        assert last.tobytes() == allc[bounces - 1].tobytes(), (what, fused)
        assert pc == rc, (what, fused, pc, rc)
        assert tot == {"rays": sum(c["rays"] for c in rc), "hits": sum(c["hits"] for c in rc)}, (what, fused)
        # the last cast alone (no events_all: the intermediate events are never written)
        _, last2, pc2, _ = bounce_on_device(g, rays, bounces, excl1, excl2, all_casts=False)
        assert last2.tobytes() == ref[bounces - 1].tobytes() and pc2 == rc, (what, fused)
    g.set_option("bounce_fused", 0)


def test_fused_bounce_loop_equals_the_oracle_cast_by_cast_in_a_closed_room():
    m = H.scenes.hall()
    T, To = H.Topology(m.verts, m.nverts), po.Topology(m.verts, m.nverts)
    rays = H.scenes.burst_rays(150_000, m.size)
    for D in (64, 128, 24):                    # one occupancy bit per voxel; the coarse bitmap (hare_voxel_bounce_tri_g); a small grid
        g, o = H.Voxel_Grid([T], D), po.VoxelGrid([To], domain=D)
        assert g.kernel_name(len(rays)).startswith("hare_voxel_pool_tri")
        check(g, To, o, rays, 6, what=f"hall D={D}")


def test_fused_bounce_loop_in_open_soups_with_quads_exclusions_and_outside_origins():
    """Rays die (open scene), quadrilaterals (hare_voxel_bounce_quad), poly_origin1 / 2 on the first cast, origins outside the grid
    (AABB.Intersect moves them: the moved-origin scratch), batches from one ray to a few pool fills, 1 .. 16 casts."""
    v, nv, size = soup(n_tri=700, n_quad=300, seed=21)
    T, To = H.Topology(v, nv), po.Topology(v, nv)
    g, o = H.Voxel_Grid([T], 12), po.VoxelGrid([To], domain=12)
    rng = np.random.default_rng(5)
    for n, B in ((1, 4), (5, 16), (64, 3), (1000, 8), (60_000, 5), (400_000, 3)):
        rays = soup_rays(n, size, seed=100 + n)
        e1 = rng.integers(-1, len(nv), n).astype(np.int32)
        e2 = rng.integers(-1, len(nv), n).astype(np.int32)
        check(g, To, o, rays, B, what=f"soup n={n} B={B}")
        check(g, To, o, rays, B, excl1=e1, excl2=e2, what=f"soup n={n} B={B} excl")
    # the same single launch behind hare_bounce_batch (host buffers; the last cast's events): one synchronisation per call
    rays = soup_rays(30_000, size, seed=77)
    ref, rc = oracle_bounce_loop(po, To, o, rays, 6)
    g.set_option("bounce_fused", 1)
    assert g.bounce_kernel_name(len(rays), 6) == "hare_voxel_bounce_quad"
    ev, c, pcs = g.Bounce_batch(rays, 6, per_cast=True)
    g.set_option("bounce_fused", 0)
    assert g.bounce_kernel_name(len(rays), 6) == ""
    ev0, c0, pcs0 = g.Bounce_batch(rays, 6, per_cast=True)
    assert ev.tobytes() == ref[5].tobytes() and ev0.tobytes() == ref[5].tobytes()
    assert [(p["rays"], p["hits"]) for p in pcs] == [(p["rays"], p["hits"]) for p in rc] == [(p["rays"], p["hits"]) for p in pcs0]
    assert (c["rays"], c["hits"]) == (c0["rays"], c0["hits"]) == (sum(p["rays"] for p in rc), sum(p["hits"] for p in rc))
    # a closed box of quads and triangles: nothing dies, every ray runs all 16 casts
    m = H.scenes.shoebox()
    Tb, Tob = H.Topology(m.verts, m.nverts), po.Topology(m.verts, m.nverts)
    gb, ob = H.Voxel_Grid([Tb], 8), po.VoxelGrid([Tob], domain=8)
    check(gb, Tob, ob, H.scenes.random_rays(20_000, m.size), 16, what="shoebox 16 casts")


def test_more_casts_than_the_fused_kernel_takes_and_the_trees_fall_back_to_a_launch_per_cast():
    m = H.scenes.shoebox()
    T, To = H.Topology(m.verts, m.nverts), po.Topology(m.verts, m.nverts)
    rays = H.scenes.random_rays(5_000, m.size)
    g, o = H.Voxel_Grid([T], 8), po.VoxelGrid([To], domain=8)
    ref, rc = oracle_bounce_loop(po, To, o, rays, 20)
    g.set_option("bounce_fused", 1)
    allc, last, pc, tot = bounce_on_device(g, rays, 20)                     # 20 > 16 casts: launch per cast even where the single launch is asked for
    assert allc.tobytes() == ref.tobytes() and pc == rc
    # counters are ACCUMULATED, per cast and in total, on both paths: two calls into the same blocks give twice the counts
    import torch
    for fused in (1, 0):
        g.set_option("bounce_fused", fused)
        ref5, rc5 = oracle_bounce_loop(po, To, o, rays, 5)
        d_ctr = torch.zeros(8, dtype=torch.int64, device="cuda")
        d_pc = torch.zeros(8 * 5, dtype=torch.int64, device="cuda")
        d_work = torch.zeros(2 * len(rays), dtype=torch.int32, device="cuda")
        d_last = torch.zeros(len(rays) * 56, dtype=torch.uint8, device="cuda")
        for _ in range(2):
            d_r = torch.from_numpy(rays).cuda()
            g.bounce_device(len(rays), d_r.data_ptr(), 5, d_work.data_ptr(), d_events_last=d_last.data_ptr(), d_counters=d_ctr.data_ptr(),
                            d_counters_per_cast=d_pc.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        pc = d_pc.cpu().numpy().reshape(5, 8)
        assert [(int(r[0]), int(r[1])) for r in pc] == [(2 * c["rays"], 2 * c["hits"]) for c in rc5], fused
        assert (int(d_ctr[0]), int(d_ctr[1])) == (2 * sum(c["rays"] for c in rc5), 2 * sum(c["hits"] for c in rc5)), fused
    g.set_option("bounce_fused", 0)
    for part, orc in ((H.Octree([T], 4, 8), po.Octree([To], 4, 8)), (H.KDTree([T], 6, 8), po.KDTree([To], 6, 8))):
        ref, rc = oracle_bounce_loop(po, To, orc, rays, 5)
        allc, last, pc, tot = bounce_on_device(part, rays, 5)
        assert allc.tobytes() == ref.tobytes() and pc == rc and last.tobytes() == ref[4].tobytes()


def test_bounce_device_argument_errors():
    import torch
    m = H.scenes.shoebox()
    g = H.Voxel_Grid([H.Topology(m.verts, m.nverts)], 8)
    rays = torch.from_numpy(H.scenes.random_rays(100, m.size)).cuda()
    work = torch.zeros(200, dtype=torch.int32, device="cuda")
    ev = torch.zeros(100 * 56, dtype=torch.uint8, device="cuda")
    with pytest.raises(H.HareError):                                        # no events at all
        g.bounce_device(100, rays.data_ptr(), 3, work.data_ptr())
    with pytest.raises(H.HareError):                                        # the work array is the ray array
        g.bounce_device(100, rays.data_ptr(), 3, rays.data_ptr(), d_events_last=ev.data_ptr())
    with pytest.raises(H.HareError):
        g.bounce_device(100, rays.data_ptr(), 0, work.data_ptr(), d_events_last=ev.data_ptr())
    g.bounce_device(0, 0, 3, 0, d_events_last=0)                            # nothing to do is not an error
