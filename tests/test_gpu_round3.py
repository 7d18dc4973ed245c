"""Round-3 GPU tests: the launch-slot ring (no memset / reduce kernel around a launch; more than 64 launches in flight),
buffer-overlap rejection, per-scene options."""
import ctypes as C

import numpy as np
import pytest

import hare_amd as H
from hare_amd import capi
from oracle import pyoracle as po
from tests.helpers import assert_events_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hall():
    m = H.scenes.hall()
    return m, H.Topology(m.verts, m.nverts), po.Topology(m.verts, m.nverts)


@pytest.mark.parametrize("kind", ["voxel_persist", "voxel_pool", "octree_persist", "octree_pool", "octree_group", "octree_dense"])
def test_more_launches_in_flight_than_launch_slots(hall, kind):
    """A scene keeps the scratch of its persistent launches (ticket word, done counters, counter shards) in a ring of 64
    slots which the launches themselves leave zeroed.  200 launches are queued here without a host synchronisation, round
    robin over five streams: the 65th must be ordered behind the first (it shares its slot), every launch must find its slot
    zeroed, and the counters the launches' last waves add up must be exact."""
    import torch
    m, T, To = hall
    if kind.startswith("voxel"):
        g, o = H.Voxel_Grid([T], 64), po.VoxelGrid([To], domain=64)
        g.set_option("voxel_kernel", 2 if kind.endswith("pool") else 1)
    else:
        g, o = H.Octree([T], 8, 16), po.Octree([To], 8, 16)
        g.set_option("octree_kernel", {"pool": 2, "group": 3, "dense": 4}.get(kind.split("_")[1], 1))
    n = 20_000
    L = 200 if kind.startswith("voxel") else 80
    rays = H.scenes.burst_rays(n, m.size)
    ref, rc = o.shoot(rays, nthreads=8)
    streams = [torch.cuda.Stream() for _ in range(5)]
    d_rays = torch.from_numpy(rays).cuda()
    outs = [torch.zeros(n * 56, dtype=torch.uint8, device="cuda") for _ in range(L)]
    ctrs = torch.zeros((L, 8), dtype=torch.int64, device="cuda")
    total = torch.zeros(8, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    for k in range(L):
        st = streams[k % len(streams)]
        g.shoot_device(n, d_rays.data_ptr(), outs[k].data_ptr(), d_counters=ctrs[k].data_ptr(), stream=st.cuda_stream)
        g.shoot_device(n, d_rays.data_ptr(), outs[k].data_ptr(), d_counters=total.data_ptr(), stream=st.cuda_stream)
    torch.cuda.synchronize()
    c = ctrs.cpu().numpy()
    assert (c[:, 0] == n).all() and (c[:, 1] == rc["hits"]).all()
    assert (int(total[0]), int(total[1])) == (L * n, L * rc["hits"])          # counters ACCUMULATE, also across streams
    for k in (0, 1, 63, 64, 65, L - 1):
        got = np.frombuffer(outs[k].cpu().numpy().tobytes(), dtype=capi.XEVENT_DTYPE)
        assert_events_equal(got, ref, what=f"{kind}, launch {k}")


def test_overlapping_device_buffers_are_rejected(hall):
    """A live ray's own X_Event slot is its scratch in the pool kernels and rays are re-read while events are written: a call
    whose buffers alias would give wrong results silently -- it is refused instead."""
    import torch
    m, T, _ = hall
    g = H.Voxel_Grid([T], 16)
    n = 1000
    buf = torch.zeros(n * 56 + n * 48, dtype=torch.uint8, device="cuda")
    ctr = torch.zeros(8, dtype=torch.int64, device="cuda")
    base = buf.data_ptr()
    with pytest.raises(H.HareError) as e:
        g.shoot_device(n, base, base + 8)                            # events on top of the rays
    assert e.value.code == capi.HARE_E_INVALID and "overlap" in str(e.value)
    with pytest.raises(H.HareError):
        g.shoot_device(n, base, base + n * 48 - 8)                   # tail of the rays under the first event
    with pytest.raises(H.HareError):
        g.shoot_device(n, base, base + n * 48, d_counters=base + n * 48 + 56)    # counters inside the events
    g.shoot_device(n, base, base + n * 48, d_counters=ctr.data_ptr())            # adjacent is fine
    torch.cuda.synchronize()
    assert int(ctr[0]) == n


def test_environment_overrides_need_the_opt_in(hall, monkeypatch):
    """HARE_VOXEL_KERNEL & co. are read once, when a scene is created, and only in a process that set HARE_DEV=1: a stray
    variable in a production environment changes nothing."""
    m, T, _ = hall
    monkeypatch.delenv("HARE_DEV", raising=False)
    monkeypatch.setenv("HARE_VOXEL_KERNEL", "persist")
    g = H.Voxel_Grid([T], 64)
    assert g.kernel_name(1000) == "hare_voxel_pool_tri"               # the library's rule (K1q), not the stray variable
    monkeypatch.setenv("HARE_DEV", "1")
    assert g.kernel_name(1000) == "hare_voxel_pool_tri"               # not re-read by an existing scene
    g2 = H.Voxel_Grid([T], 64)
    assert g2.kernel_name(1000) == "hare_voxel_persist_tri"
    monkeypatch.setenv("HARE_VOXEL_KERNEL", "pool")
    assert g2.kernel_name(1000) == "hare_voxel_persist_tri"


# ---------------------------------------------------------------------------------------------------------------
# hare_bounce_batch: the whole bounce loop behind one C-ABI call from host buffers
from tests.helpers import oracle_bounce_loop, soup, soup_rays   # noqa: E402


def test_bounce_batch_c5_full_size_every_cast_equals_the_oracle():
    """BASELINE config[4] at its per-GPU size through ONE call: 1 048 576 rays x 8 casts in the 1M-triangle cathedral, D = 128;
    the events of every cast, the summed and the per-cast counters equal the oracle's loop."""
    m = H.scenes.cathedral()
    T, To = H.Topology(m.verts, m.nverts), po.Topology(m.verts, m.nverts)
    n, B = 1 << 20, 8
    rays = H.scenes.burst_rays(8 << 20, m.size, start=5 << 20, count=n)     # one rank's shard of the 8M-ray burst
    g, o = H.Voxel_Grid([T], 128), po.VoxelGrid([To], domain=128)
    ref, rc = oracle_bounce_loop(po, To, o, rays, B)
    ev, c, pcs = g.Bounce_batch(rays, B, all_casts=True, per_cast=True)
    for b in range(B):
        assert_events_equal(ev[b], ref[b], what=f"bounce batch, cast {b}")
        assert (pcs[b]["rays"], pcs[b]["hits"]) == (rc[b]["rays"], rc[b]["hits"])
    assert (c["rays"], c["hits"]) == (sum(x["rays"] for x in rc), sum(x["hits"] for x in rc))
    last, c2 = g.Bounce_batch(rays, B)                                        # events of the last cast only
    assert_events_equal(last, ref[B - 1], what="bounce batch, last cast only")
    assert c2 == c


@pytest.mark.parametrize("kind", ["voxel_persist", "voxel_pool", "octree", "kdtree"])
def test_bounce_batch_open_scene_packs_the_survivors(kind):
    """A polygon soup with no walls: most rays leave within a few bounces.  The loop packs the survivors (stably) once a
    quarter of the rays in flight have died and later reflects in place again -- every cast still equals the oracle's, in
    the caller's ray order, with quads, caller exclusions on the first cast and origins outside the grid."""
    v, nv, size = soup(n_tri=1500, n_quad=400, seed=5)
    T, To = H.Topology(v, nv), po.Topology(v, nv)
    n, B = (60_000, 7) if kind != "kdtree" else (20_000, 4)
    rays = soup_rays(n, size, seed=21)
    rng = np.random.default_rng(2)
    e1 = rng.integers(-3, len(nv), n).astype(np.int32)       # negative: excludes nothing (the reference compares indices only)
    e2 = rng.integers(-1, len(nv), n).astype(np.int32)
    if kind.startswith("voxel"):
        g, o = H.Voxel_Grid([T], 24), po.VoxelGrid([To], domain=24)
        g.set_option("voxel_kernel", 2 if kind.endswith("pool") else 1)
    elif kind == "octree":
        g, o = H.Octree([T], 5, 8), po.Octree([To], 5, 8)
    else:
        g, o = H.KDTree([T], 8, 8), po.KDTree([To], 8, 8)
    ref, rc = oracle_bounce_loop(po, To, o, rays, B, e1, e2)
    ev, c, pcs = g.Bounce_batch(rays, B, poly_origin1=e1, poly_origin2=e2, all_casts=True, per_cast=True)
    alive = [x["rays"] for x in rc]
    assert alive[0] == n and alive[2] < 0.75 * n and alive[-1] > 0, alive       # the packing rule did trigger, rays were left
    for b in range(B):
        assert_events_equal(ev[b], ref[b], what=f"{kind}: open scene, cast {b}")
        assert (pcs[b]["rays"], pcs[b]["hits"]) == (rc[b]["rays"], rc[b]["hits"]), (b, pcs[b], rc[b])
    assert c["rays"] == sum(alive)


def test_bounce_batch_edges_and_sharding():
    m = H.scenes.shoebox()
    T, To = H.Topology(m.verts, m.nverts), po.Topology(m.verts, m.nverts)
    g, o = H.Voxel_Grid([T], 8), po.VoxelGrid([To], domain=8)
    rays = H.scenes.random_rays(5000, m.size)
    # one cast == Shoot_batch
    ev1, c1 = g.Bounce_batch(rays, 1)
    ev0, c0 = g.Shoot_batch(rays)
    assert ev1.tobytes() == ev0.tobytes() and (c1["rays"], c1["hits"]) == (c0["rays"], c0["hits"])
    # no ray, one ray, every ray dead after the first cast (origins far outside, pointing away)
    ev, c = g.Bounce_batch(np.zeros((0, 6)), 3, all_casts=True)
    assert ev.shape == (3, 0) and c["rays"] == 0
    ev, c = g.Bounce_batch(rays[:1], 4, all_casts=True)
    ref, _ = oracle_bounce_loop(po, To, o, rays[:1], 4)
    assert ev.tobytes() == ref.tobytes()
    away = rays.copy()
    away[:, :3] = 100.0
    away[:, 3:] = 1.0
    ev, c, pcs = g.Bounce_batch(away, 5, all_casts=True, per_cast=True)
    assert not ev["hit"].any() and (ev["poly_id"] == -1).all() and c["rays"] == len(away) and pcs[1]["rays"] == 0
    # two scenes, the second on device 1 where the box has one (else both on this device): byte-identical to the one-scene call
    g2 = H.Voxel_Grid([T], 8, device=1 % H.device_count())
    ref, rc = oracle_bounce_loop(po, To, o, rays, 6)
    a, ca = g.Bounce_batch(rays, 6, all_casts=True)
    b, cb = H.Spatial_Partition.Bounce_batch_sharded([g, g2], rays, 6, all_casts=True)
    assert a.tobytes() == ref.tobytes() and b.tobytes() == ref.tobytes() and ca == cb
    with pytest.raises(H.HareError):
        g.Bounce_batch(rays, 0)


# ---------------------------------------------------------------------------------------------------------------
# The occlusion predicate without events: traversal ends as soon as a ray's flag is decided
def _want(ref, tmax):
    return ((ref["hit"] != 0) & (True if tmax is None else (ref["t"] < tmax))).astype(np.int32)


@pytest.mark.parametrize("domain", [64, 128, 9])
def test_bounded_occlusion_equals_the_closest_hit_predicate_voxel(hall, domain):
    """t_max distributions from far too short to far too long, scaled to every ray's own hit distance and to the room's mean free
    path; exactly at the hit (strict <); infinities, NaN, negatives; rays that start outside the grid (t includes the clip
    distance).  Flags from the flags-only kernels == flags derived from the oracle's closest hit, on every ray."""
    import torch
    m, T, To = hall
    n = 400_000
    rays = H.scenes.burst_rays(n, m.size)
    rng = np.random.default_rng(7)
    rays[::5, :3] += rng.normal(0, 30.0, (len(rays[::5]), 3))           # a fifth of the origins far outside the grid
    rays[1::5, :3] = rng.uniform(0, 1, (len(rays[1::5]), 3)) * np.asarray(m.size)
    o = po.VoxelGrid([To], domain=domain)
    ref, _ = o.shoot(rays, nthreads=16)
    g = H.Voxel_Grid([T], domain)
    mfp = float(np.median(ref["t"][ref["hit"] != 0]))
    scale = rng.choice([0.02, 0.1, 0.25, 0.5, 0.9, 0.999999, 1.0, 1.000001, 1.1, 2.0, 10.0], n)
    tmax = np.where(rng.random(n) < 0.5, ref["t"] * scale, mfp * scale * rng.random(n) * 2)
    tmax[::101] = np.inf
    tmax[7::101] = np.nan
    tmax[13::101] = -1.0
    tmax[19::101] = 0.0
    want = _want(ref, tmax)
    assert 0.2 < want.mean() < 0.8
    assert g.kernel_name(n) .startswith("hare_voxel_p")                    # Shoot itself is untouched
    occ, c = g.Occluded_batch(rays, tmax, events=False)
    bad = np.nonzero(occ != want)[0]
    assert bad.size == 0, (domain, bad[:5], tmax[bad[:5]], ref[bad[:5]])
    assert (c["rays"], c["hits"]) == (n, int(want.sum()))
    occ_any, _ = g.Occluded_batch(rays, None, events=False)               # no t_max: any hit
    assert np.array_equal(occ_any, _want(ref, None))
    occ_e, ev = g.Occluded_batch(rays, tmax)                               # with events: the closest-hit cast, identical flags
    assert np.array_equal(occ_e, want) and ev.tobytes() == ref.tobytes()
    occ_s, _ = g.Occluded_batch(rays[:50_000], tmax[:50_000], events=False, simple_kernel=True)
    assert np.array_equal(occ_s, want[:50_000])
    # device-resident form, no events buffer
    d_rays = torch.from_numpy(rays).cuda()
    d_tmax = torch.from_numpy(tmax).cuda()
    d_occ = torch.full((n,), 7, dtype=torch.int32, device="cuda")
    d_ctr = torch.zeros(8, dtype=torch.int64, device="cuda")
    g.occluded_device(n, d_rays.data_ptr(), 0, d_occ.data_ptr(), d_tmax=d_tmax.data_ptr(), d_counters=d_ctr.data_ptr(),
                      stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(d_occ.cpu().numpy(), want) and int(d_ctr[1]) == int(want.sum())
    assert np.array_equal(d_rays.cpu().numpy(), rays)                     # a predicate: the rays are not touched


def test_bounded_occlusion_quads_exclusions_and_trees():
    v, nv, size = soup(n_tri=1500, n_quad=400, seed=9)
    T, To = H.Topology(v, nv), po.Topology(v, nv)
    n = 80_000
    rays = soup_rays(n, size, seed=31)
    rng = np.random.default_rng(3)
    e1 = rng.integers(-3, len(nv), n).astype(np.int32)
    e2 = rng.integers(-1, len(nv), n).astype(np.int32)
    for name, g, o in (("voxel", H.Voxel_Grid([T], 20), po.VoxelGrid([To], domain=20)),
                       ("octree", H.Octree([T], 5, 8), po.Octree([To], 5, 8)),
                       ("kdtree", H.KDTree([T], 8, 8), po.KDTree([To], 8, 8))):
        m_ = n if name != "kdtree" else 15_000
        ref, _ = o.shoot(rays[:m_], excl1=e1[:m_], excl2=e2[:m_], nthreads=16)
        tmax = ref["t"] * rng.choice([0.3, 0.999999, 1.0, 1.000001, 3.0], m_)
        tmax[ref["hit"] == 0] = rng.uniform(0.1, 20.0, int((ref["hit"] == 0).sum()))
        want = _want(ref, tmax)
        assert 0.02 * m_ < want.sum() < 0.9 * m_, (name, want.mean())      # an open soup: most rays hit nothing
        occ, c = g.Occluded_batch(rays[:m_], tmax, poly_origin1=e1[:m_], poly_origin2=e2[:m_], events=False)
        bad = np.nonzero(occ != want)[0]
        assert bad.size == 0, (name, bad[:5])
        assert c["hits"] == int(want.sum())
        occ_any, _ = g.Occluded_batch(rays[:m_], None, poly_origin1=e1[:m_], poly_origin2=e2[:m_], events=False)
        assert np.array_equal(occ_any, _want(ref, None)), name


def test_bounded_octree_occlusion_on_the_hall(hall):
    """The octree's closest hit is the reference's, quirk included (far children first, early return: DESIGN.md F15) -- the
    flags-only kernel stops at the first hit below t_max and must still agree with it on every ray."""
    m, T, To = hall
    n = 200_000
    rays = H.scenes.burst_rays(n, m.size)
    refo, _ = po.Octree([To], 8, 16).shoot(rays, nthreads=16)
    rng = np.random.default_rng(12)
    tmax = refo["t"] * rng.choice([0.1, 0.5, 0.999999, 1.0, 1.000001, 1.5, 4.0], n)
    oc = H.Octree([T], 8, 16)
    occ, c = oc.Occluded_batch(rays, tmax, events=False)
    want = _want(refo, tmax)
    assert np.array_equal(occ, want) and c["hits"] == int(want.sum())
    # no t_max at all: any hit decides (round 5: hare_octree_occl_any, K2p's OCC build; with a t_max array the dense build, hare_octree_occl)
    occ_any, c_any = oc.Occluded_batch(rays, None, events=False)
    assert np.array_equal(occ_any, (refo["hit"] == 1).astype(np.int32)) and c_any["hits"] == int((refo["hit"] == 1).sum())
    # ... and the same flags with exclusions, both builds
    e1 = refo["poly_id"].astype(np.int32).copy(); e1[::3] = -1
    refx, _ = po.Octree([To], 8, 16).shoot(rays[:50_000], excl1=e1[:50_000], nthreads=16)
    occ_x, _ = oc.Occluded_batch(rays[:50_000], tmax[:50_000], poly_origin1=e1[:50_000], events=False)
    assert np.array_equal(occ_x, _want(refx, tmax[:50_000]))
    occ_xa, _ = oc.Occluded_batch(rays[:50_000], None, poly_origin1=e1[:50_000], events=False)
    assert np.array_equal(occ_xa, (refx["hit"] == 1).astype(np.int32))


# ---------------------------------------------------------------------------------------------------------------
# Slim result records (HARE_SHOOT_SLIM_EVENTS): 16 / 32 bytes per ray over the host link, X_Events rebuilt bit for bit
@pytest.mark.parametrize("kernel", [0, 1, 2])
def test_slim_events_rebuild_the_full_records_voxel(hall, kernel):
    m, T, To = hall
    g = H.Voxel_Grid([T], 64)
    g.set_option("voxel_kernel", kernel)
    n = 300_000
    rays = H.scenes.burst_rays(n, m.size)
    rng = np.random.default_rng(17)
    rays[::4, :3] += rng.normal(0, 40.0, (len(rays[::4]), 3))              # a quarter start outside the grid: AABB.Intersect moves them
    full, c = g.Shoot_batch(rays)
    slim, c2 = g.Shoot_batch(rays, slim=True)
    assert slim.dtype.itemsize == 16 and c2 == c
    assert (slim["hit"] == 2).sum() > 1000 and (slim["hit"] == 1).sum() > 1000 and (slim["hit"] == 0).sum() > 1000
    back = g.expand_events(rays, slim)
    assert back.tobytes() == full.tobytes()
    # the one-line rebuild a caller can do itself for rays that start inside the grid (hit == 1): X_Point = o + d * t
    k = slim["hit"] == 1
    for a, col in enumerate(("x", "y", "z")):
        assert np.array_equal(rays[k, a] + rays[k, 3 + a] * slim["t"][k], full[col][k])
    assert np.array_equal(slim["t"][k], full["t"][k]) and np.array_equal(slim["poly_id"], full["poly_id"])
    # the oracle agrees with the rebuilt records (so slim is pinned by the same checker)
    ref, _ = po.VoxelGrid([To], domain=64).shoot(rays, nthreads=16)
    assert_events_equal(back, ref, what="slim -> expanded events")


def test_slim_events_trees_quads_and_sharding():
    v, nv, size = soup(n_tri=800, n_quad=300, seed=4)
    T = H.Topology(v, nv)
    rays = soup_rays(40_000, size, seed=8)
    for g in (H.Voxel_Grid([T], 14), H.Octree([T], 5, 8), H.KDTree([T], 8, 8)):
        r = rays if g._kind != capi.KIND_KDTREE else rays[:8000]
        full, c = g.Shoot_batch(r)
        slim, c2 = g.Shoot_batch(r, slim=True)
        assert slim.dtype.itemsize == (16 if g._kind == capi.KIND_VOXEL else 32) and c2 == c
        assert g.expand_events(r, slim).tobytes() == full.tobytes()
    g1, g2 = H.Octree([T], 5, 8), H.Octree([T], 5, 8, device=1 % H.device_count())
    full, _ = g1.Shoot_batch(rays)
    slim, _ = H.Spatial_Partition.Shoot_batch_sharded([g1, g2], rays, slim=True)
    assert g1.expand_events(rays, slim).tobytes() == full.tobytes()
    m = H.scenes.shoebox()                                                 # config 1: the committed golden set's scene
    gs = H.Voxel_Grid([H.Topology(m.verts, m.nverts)], 8)
    rr = H.scenes.random_rays(10_000, m.size)
    full, _ = gs.Shoot_batch(rr)
    assert gs.expand_events(rr, gs.Shoot_batch(rr, slim=True)[0]).tobytes() == full.tobytes()


# ---------------------------------------------------------------------------------------------------------------
# The cooperative tail (voxel_coop.hip): a drained wave traces its last rays with all 64 lanes
@pytest.mark.parametrize("kernel", [1, 2])
def test_cooperative_tail_is_invisible_in_the_results(hall, kernel):
    """Rays that skim the hall's floor, walls and displaced ceiling cross a hundred occupied voxels and scan thousands of list
    entries: the rays a wave is left with at the end of a launch, which it then traces cooperatively (64 lanes on one voxel's
    list, chunk minima with the earlier entry winning ties).  Results with the tail on, off, and from the oracle must be the same
    bytes -- also with exclusions, with origins outside the grid, and in a batch so small that every wave goes cooperative at once."""
    m, T, To = hall
    rng = np.random.default_rng(23)
    n = 120_000
    L = np.asarray(m.size)
    o = rng.uniform(0.02, 0.98, (n, 3)) * L
    d = rng.normal(size=(n, 3))
    axis = rng.integers(0, 3, n)
    d[np.arange(n), axis] *= 1e-3                                   # nearly parallel to a pair of walls ...
    o[np.arange(n), axis] = np.where(rng.random(n) < 0.5, 0.004, L[axis] - 0.004) + rng.normal(0, 1e-3, n)   # ... and millimetres from one of them
    o[::9] += rng.normal(0, 25.0, (len(o[::9]), 3))                 # some start far outside the grid
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.ascontiguousarray(np.concatenate([o, d], axis=1))
    e1 = rng.integers(-1, m.P, n).astype(np.int32)
    ref, rc = po.VoxelGrid([To], domain=64).shoot(rays, excl1=e1, nthreads=16)
    assert rc["entries"] / n > 150                                  # heavy rays indeed (the burst scans 18 entries per ray)
    g = H.Voxel_Grid([T], 64)
    g.set_option("voxel_kernel", kernel)
    got = {}
    for coop in (1, 0):
        g.set_option("coop_tail", coop)
        got[coop], c = g.Shoot_batch(rays, poly_origin1=e1)
        assert_events_equal(got[coop], ref, what=f"kernel {kernel}, coop_tail {coop}")
        assert (c["rays"], c["hits"]) == (n, rc["hits"])
    assert got[0].tobytes() == got[1].tobytes()
    g.set_option("coop_tail", 1)
    for k in (1, 3, 64, 700):                                       # launches in which a wave holds only a few rays from the start
        ev, _ = g.Shoot_batch(rays[:k], poly_origin1=e1[:k])
        assert ev.tobytes() == ref[:k].tobytes()


@pytest.mark.parametrize("domain", [64, 128])
def test_wide_drain_modes_are_invisible_in_the_results(hall, domain):
    """K1q's drain (voxel_pool.hip): once the tickets are dry and at most 64 rays are left in a wave's pool, the pre-cull runs
    WIDE (a ray's candidates four per lane over 1 - 16 lanes, first survivor in list order) and the walk looks several occupied
    voxels ahead (one per lane, whole lists).  Results with the modes on, off and from the oracle must be the same bytes: on the
    burst, on surface-skimming rays (long lists, a hundred occupied voxels), with both exclusions, with origins outside the grid
    (moved origins: t_start), with origin write-back, on a coarse bitmap (D = 128) and in launches of every size from one ray to
    a few per wave -- where a wave is in its drain from the first round."""
    m, T, To = hall
    rng = np.random.default_rng(41)
    n = 150_000
    L = np.asarray(m.size)
    o = rng.uniform(0.02, 0.98, (n, 3)) * L
    d = rng.normal(size=(n, 3))
    axis = rng.integers(0, 3, n)
    skim = rng.random(n) < 0.5
    d[skim, axis[skim]] *= 2e-3
    o[skim, axis[skim]] = np.where(rng.random(skim.sum()) < 0.5, 0.006, L[axis[skim]] - 0.006) + rng.normal(0, 2e-3, skim.sum())
    o[::7] += rng.normal(0, 30.0, (len(o[::7]), 3))                 # origins far outside: AABB.Intersect moves them (or they miss)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.ascontiguousarray(np.concatenate([o, d], axis=1))
    e1 = rng.integers(-1, m.P, n).astype(np.int32)
    e2 = rng.integers(-1, m.P, n).astype(np.int32)
    og = po.VoxelGrid([To], domain=domain)
    ref, rc = og.shoot(rays, excl1=e1, excl2=e2, nthreads=16)
    g = H.Voxel_Grid([T], domain)
    g.set_option("voxel_kernel", 2)
    got = {}
    for wide in (1, 0):
        g.set_option("wide_drain", wide)
        got[wide], c = g.Shoot_batch(rays, poly_origin1=e1, poly_origin2=e2)
        assert_events_equal(got[wide], ref, what=f"D={domain}, wide_drain {wide}")
        assert (c["rays"], c["hits"]) == (n, rc["hits"])
    assert got[0].tobytes() == got[1].tobytes()
    g.set_option("wide_drain", 1)
    for k in (1, 2, 5, 33, 64, 65, 1000, 3072 * 3):                 # a wave holds a few rays from its first round on
        ev, _ = g.Shoot_batch(rays[:k], poly_origin1=e1[:k], poly_origin2=e2[:k])
        assert ev.tobytes() == ref[:k].tobytes(), k
    # origin write-back: the moved origins come back in rays[], the events are the same (the cooperative tail is off then, the wide modes are not)
    r1 = rays[:40_000].copy()
    ev, _ = g.Shoot_batch(r1, poly_origin1=e1[:40_000], poly_origin2=e2[:40_000], writeback_origin=True)
    refw, _, moved = og.shoot(rays[:40_000], excl1=e1[:40_000], excl2=e2[:40_000], mutate=True, nthreads=16)
    assert_events_equal(ev, refw, what="write-back")
    assert np.array_equal(r1, moved)


def test_wide_drain_modes_with_quadrilaterals():
    """The same on a soup with quadrilaterals (never culled: every quad survives the wide cull and goes to the exact phase)."""
    from tests.helpers import soup, soup_rays
    v, nv, size = soup(n_tri=3000, n_quad=800, seed=5)
    rays = soup_rays(60_000, size, seed=12)
    rng = np.random.default_rng(2)
    e1 = rng.integers(-1, len(nv), len(rays)).astype(np.int32)
    ref, rc = po.VoxelGrid([po.Topology(v, nv)], domain=10).shoot(rays, excl1=e1, nthreads=16)
    g = H.Voxel_Grid([H.Topology(v, nv)], 10)
    g.set_option("voxel_kernel", 2)
    out = {}
    for wide in (1, 0):
        g.set_option("wide_drain", wide)
        out[wide], c = g.Shoot_batch(rays, poly_origin1=e1)
        assert_events_equal(out[wide], ref, what=f"quads, wide_drain {wide}")
        assert c["hits"] == rc["hits"]
    assert out[0].tobytes() == out[1].tobytes()


def test_octree_tail_kernel_is_invisible_in_the_results(hall):
    """K2p hands the rays its waves still walk at the end of a launch to a tail kernel: K2g-tail (octree_group.hip: eight lanes per ray,
    continued from K2p's frames -- the default) or K2t (octree_coop.hip: a wave per ray; the children of a frame on eight
    lanes, a leaf's list replayed from per-lane results with the reference's strict-< scan and its early return).  With the hand-over
    on, off, and from the oracle the events must be the same bytes: burst rays, surface-skimming rays (the heavy ones), exclusions,
    a deep tree over a small crowded scene, and batches so small that everything is handed over."""
    m, T, To = hall
    rng = np.random.default_rng(29)
    n = 150_000
    L = np.asarray(m.size)
    o = rng.uniform(0.02, 0.98, (n, 3)) * L
    d = rng.normal(size=(n, 3))
    axis = rng.integers(0, 3, n)
    half = n // 2
    d[np.arange(half), axis[:half]] *= 1e-3                          # half of them skim a wall, the floor or the ceiling
    o[np.arange(half), axis[:half]] = np.where(rng.random(half) < 0.5, 0.004, L[axis[:half]] - 0.004)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.ascontiguousarray(np.concatenate([o, d], axis=1))
    rays[half:] = H.scenes.burst_rays(n - half, m.size)
    e1 = rng.integers(-1, m.P, n).astype(np.int32)
    oc, oo = H.Octree([T], 8, 16), po.Octree([To], 8, 16)
    oc.set_option("octree_kernel", 1)                                 # K2p + its tail kernels: what this test is about
    assert oc.kernel_name(n) == "hare_octree_persist"
    ref, rc = oo.shoot(rays, excl1=e1, nthreads=16)
    got = {}
    # no tail (every K2p lane finishes its own ray), K2t (a wave per ray: a wave's last 16 rays after 64 rounds), K2g-tail (eight lanes
    # per ray: every ray a wave still walks 32 rounds after its tickets ran dry -- the default), and K2g-tail taking ALL rays at once
    for tail, extra in ((0, {}), (1, {}), (2, {}), (2, {"k2p_tail_max": 64, "k2p_tail_patience": 0}), (2, {"k2p_tail_max": 7, "k2p_tail_patience": 3})):
        oc.set_option("octree_tail", tail)
        for k_, v_ in {"k2p_tail_max": 0, "k2p_tail_patience": -1, **extra}.items():
            oc.set_option(k_, v_)
        key = (tail, tuple(sorted(extra.items())))
        got[key], c = oc.Shoot_batch(rays, poly_origin1=e1)
        assert_events_equal(got[key], ref, what=f"octree, tail kernel {key}")
        assert (c["rays"], c["hits"]) == (n, rc["hits"]), key
    assert len({g_.tobytes() for g_ in got.values()}) == 1
    oc.set_option("coop_tail", 0)                                     # the scene-wide switch turns every tail off as well
    assert oc.Shoot_batch(rays, poly_origin1=e1)[0].tobytes() == got[(0, ())].tobytes()
    oc.set_option("coop_tail", 1)
    for tail in (1, 2):
        oc.set_option("octree_tail", tail)
        oc.set_option("k2p_tail_max", 64 if tail == 2 else 0)
        oc.set_option("k2p_tail_patience", 0 if tail == 2 else -1)
        for k in (1, 5, 64, 900):
            ev, c = oc.Shoot_batch(rays[:k], poly_origin1=e1[:k])
            assert ev.tobytes() == ref[:k].tobytes() and c["hits"] == int(ref["hit"][:k].sum())
    oc.set_option("k2p_tail_max", 0)
    oc.set_option("k2p_tail_patience", -1)
    v, nv, size = soup(n_tri=900, n_quad=300, seed=6)                 # quadrilaterals, a tree of 7 levels with 2 polygons per leaf
    sr = soup_rays(50_000, size, seed=14)
    g2, o2 = H.Octree([H.Topology(v, nv)], 7, 2), po.Octree([po.Topology(v, nv)], 7, 2)
    ref2 = o2.shoot(sr, nthreads=16)[0]
    for kern in (0, 1, 3):                                            # the rule (K2g at this size), K2p + K2g-tail, K2g
        g2.set_option("octree_kernel", kern)
        assert_events_equal(g2.Shoot_batch(sr)[0], ref2, what=f"octree 7/2 with quads, kernel option {kern}")
