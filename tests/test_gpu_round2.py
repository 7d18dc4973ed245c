"""GPU parity tests added in round 2: occlusion predicate (A9), the "retired ray" sentinel only in the bounce loop,
hare_shoot_one == batch kernels, deep octrees (17+ levels), config 5 at its full per-GPU size, the caller's
current device left alone."""
import ctypes as C

import numpy as np
import pytest

import hare_amd as H
from hare_amd import capi
from oracle import pyoracle as po
from tests.helpers import assert_events_equal, soup, soup_rays

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hall():
    m = H.scenes.hall()
    return m, H.Topology(m.verts, m.nverts), po.Topology(m.verts, m.nverts)


def test_occlusion_predicate_batch_and_device(hall):
    """occluded = closest hit exists and t < t_max (harness-defined) -- pinned by the closest-hit oracle."""
    import torch
    m, T, To = hall
    n = 300_000
    rays = H.scenes.burst_rays(n, m.size)
    rng = np.random.default_rng(4)
    ref, _ = po.VoxelGrid([To], domain=64).shoot(rays, nthreads=16)
    tmax = ref["t"] * rng.choice([0.5, 1.0, 1.5], n)          # before / exactly at (strict <: not occluded) / behind the hit
    tmax[::11] = np.inf
    tmax[5::11] = -1.0
    g = H.Voxel_Grid([T], 64)
    want = ((ref["hit"] != 0) & (ref["t"] < tmax)).astype(np.int32)
    assert 0.2 < want.mean() < 0.8
    occ, ev = g.Occluded_batch(rays, tmax)
    assert_events_equal(ev, ref, what="occlusion: closest-hit records")
    assert np.array_equal(occ, want)
    occ_any, _ = g.Occluded_batch(rays)                       # t_max None: any hit
    assert np.array_equal(occ_any, (ref["hit"] != 0).astype(np.int32))
    # device-resident form
    d_rays = torch.from_numpy(rays).cuda()
    d_tmax = torch.from_numpy(tmax).cuda()
    d_ev = torch.empty(n * 56, dtype=torch.uint8, device="cuda")
    d_occ = torch.full((n,), 7, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    g.occluded_device(n, d_rays.data_ptr(), d_ev.data_ptr(), d_occ.data_ptr(), d_tmax=d_tmax.data_ptr(), stream=st)
    torch.cuda.synchronize()
    assert np.array_equal(d_occ.cpu().numpy(), want)
    assert np.frombuffer(d_ev.cpu().numpy().tobytes(), dtype=capi.XEVENT_DTYPE).tobytes() == ref.tobytes()
    # octree path: same predicate on ITS closest-hit records (which differ from the grid's)
    oc = H.Octree([T], 8, 16)
    refo, _ = po.Octree([To], 8, 16).shoot(rays[:50_000], nthreads=16)
    occ, ev = oc.Occluded_batch(rays[:50_000], tmax[:50_000])
    assert_events_equal(ev, refo, what="occlusion: octree records")
    assert np.array_equal(occ, ((refo["hit"] != 0) & (refo["t"] < tmax[:50_000])).astype(np.int32))


def test_negative_poly_origin_is_no_exclusion_outside_the_bounce_loop(hall):
    """Shoot(R, 0, out e, -2) == Shoot(R, 0, out e): the reference compares indices only (Voxel_Grid.cs:477).  The -2
    'retired' mark of hare_reflect_device is honoured only with HARE_SHOOT_RETIRED_RAYS on the device-resident call."""
    import torch
    m, T, To = hall
    n = 50_000
    rays = H.scenes.burst_rays(n, m.size)
    e1 = np.full(n, -2, np.int32)
    e1[::3] = -7
    for part in (H.Voxel_Grid([T], 64), H.Octree([T], 8, 16)):
        plain, c0 = part.Shoot_batch(rays)
        neg, c1 = part.Shoot_batch(rays, poly_origin1=e1)
        assert neg.tobytes() == plain.tobytes() and c1["rays"] == n and c1["hits"] == c0["hits"]
        d_rays = torch.from_numpy(rays).cuda()
        d_e1 = torch.from_numpy(e1).cuda()
        d_ev = torch.empty(n * 56, dtype=torch.uint8, device="cuda")
        d_ctr = torch.zeros(8, dtype=torch.int64, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        part.shoot_device(n, d_rays.data_ptr(), d_ev.data_ptr(), d_excl1=d_e1.data_ptr(), d_counters=d_ctr.data_ptr(), stream=st)
        torch.cuda.synchronize()
        assert d_ev.cpu().numpy().tobytes() == plain.tobytes() and int(d_ctr[0]) == n
        d_ctr.zero_()
        part.shoot_device(n, d_rays.data_ptr(), d_ev.data_ptr(), d_excl1=d_e1.data_ptr(), d_counters=d_ctr.data_ptr(), stream=st,
                          flags=capi.SHOOT_RETIRED_RAYS)
        torch.cuda.synchronize()
        ev = np.frombuffer(d_ev.cpu().numpy().tobytes(), dtype=capi.XEVENT_DTYPE)
        dead = e1 == -2
        assert not ev["hit"][dead].any() and (ev["poly_id"][dead] == -1).all()
        assert ev[~dead].tobytes() == plain[~dead].tobytes()
        assert int(d_ctr[0]) == int((~dead).sum())            # retired rays are not counted


def test_developer_flag_bits_are_masked_without_HARE_DEV(hall, monkeypatch):
    monkeypatch.delenv("HARE_DEV", raising=False)
    m, T, _ = hall
    rays = H.scenes.burst_rays(20_000, m.size)
    g = H.Voxel_Grid([T], 64)
    plain, _ = g.Shoot_batch(rays)
    out = np.zeros(len(rays), capi.XEVENT_DTYPE)
    ctr = capi.Counters()
    for bits in (0x2000, 0x4000, 0x8000, 0xFFFFFFE0):          # everything but the five public bits (1, 2, 4, 8, 16)
        capi.check(capi.lib.hare_shoot_batch(g._h, 0, 0, len(rays), rays.ctypes.data, None, None, bits, out.ctypes.data,
                                             C.addressof(ctr)))
        assert out.tobytes() == plain.tobytes() and ctr.rays == len(rays)


def test_shoot_one_equals_the_batch_kernels(hall):
    """The host single-ray path and the HIP batch path are the same function of (scene, ray): 20k rays each way."""
    m, T, _ = hall
    rays = H.scenes.burst_rays(1 << 20, m.size)[:: (1 << 20) // 20_000].copy()
    for part in (H.Voxel_Grid([T], 64), H.Octree([T], 8, 16)):
        batch, _ = part.Shoot_batch(rays)
        one = np.zeros(len(rays), capi.XEVENT_DTYPE)
        r = rays.copy()
        for i in range(len(rays)):
            one[i] = part.Shoot_one(r[i])
        assert one.tobytes() == batch.tobytes()


def deep_scene(depth, seed=1):
    """A scene whose octree really reaches `depth` levels without exploding.  The reference pads every child box by an
    ABSOLUTE 0.1 m ("Octree - alt.cs":99-111), so node size tends to 0.4 m and below that every polygon lands in all 8
    children: a deep tree is only finite when the model is huge (here 0.4 * 2^(depth-2) m) and the crowded spots tiny."""
    rng = np.random.default_rng(seed)
    E = 0.4 * 2.0 ** (depth - 2)
    tris = [np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], float),               # pins the model's min corner
            np.array([[E, E, E], [E - 1, E, E], [E, E - 1, E]], float)]
    centres = rng.uniform(0.1, 0.9, (6, 3)) * E
    for c in centres:
        for _ in range(3):                                                      # three tiny triangles within a few cm
            tris.append(c + rng.uniform(-0.03, 0.03, 3) + rng.uniform(-0.02, 0.02, (3, 3)))
    v = np.zeros((len(tris), 4, 3))
    v[:, :3] = H.scenes.snap(np.array(tris))
    nv = np.full(len(tris), 3, np.int32)
    o = np.concatenate([centres[rng.integers(0, 6, 3000)] + rng.normal(size=(3000, 3)) * 2.0, rng.uniform(0, 1, (1000, 3)) * E])
    tgt = centres[rng.integers(0, 6, 4000)] + rng.uniform(-0.03, 0.03, (4000, 3))
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return v, nv, np.ascontiguousarray(np.concatenate([o, d], 1))


def tree_depth(fc):
    depth, level = 0, [0]
    while level:
        nxt = [c for n in level if fc[n] >= 0 for c in range(fc[n], fc[n] + 8)]
        if nxt:
            depth += 1
        level = nxt
    return depth


@pytest.mark.parametrize("depth,maxp", [(17, 1), (20, 2), (24, 1)])
def test_deep_octrees_17_to_24_levels(depth, maxp):
    """maxDepth above the 16 levels round 1's persistent kernel stopped at, on trees that really get that deep:
    persistent, simple and counting kernels vs the oracle."""
    v, nv, rays = deep_scene(depth)
    g = H.Octree([H.Topology(v, nv)], depth, maxp)
    o = po.Octree([po.Topology(v, nv)], depth, maxp)
    _, fc, _, _, _ = g.nodes()
    assert g.info().n_nodes == o.n_nodes and tree_depth(fc) == depth
    ref, rc = o.shoot(rays)
    assert 100 < ref["hit"].sum() < len(rays)
    assert_events_equal(g.Shoot_batch(rays)[0], ref, what=f"octree depth {depth} (persistent)")
    assert_events_equal(g.Shoot_batch(rays, simple_kernel=True)[0], ref, what=f"octree depth {depth} (simple)")
    ev, ctr = g.Shoot_batch(rays, count_work=True)
    assert_events_equal(ev, ref, what=f"octree depth {depth} (counting)")
    assert ctr["cells"] == rc["cells"] and ctr["entries"] == rc["entries"]


def test_octree_depth_beyond_the_supported_range_is_a_clean_error():
    v, nv, _ = soup(20, 0)
    with pytest.raises(H.HareError) as ei:
        H.Octree([H.Topology(v, nv)], 25, 1)
    assert ei.value.code == capi.HARE_E_INVALID and "max_depth" in str(ei.value)


def test_calls_leave_the_callers_current_device_alone(hall):
    """Every entry point acts on the scene's device and restores the calling thread's current device (one GPU here:
    what can be checked is that torch's current device and stream are as before, and that a scene on an ordinal
    that does not exist is refused with a code, not a crash)."""
    import torch
    m, T, _ = hall
    before = torch.cuda.current_device()
    g = H.Voxel_Grid([T], 16)
    g.Shoot_batch(H.scenes.burst_rays(1000, m.size))
    g.close()
    assert torch.cuda.current_device() == before
    with pytest.raises(H.HareError) as ei:
        H.Voxel_Grid([T], 16, device=63)
    assert ei.value.code in (capi.HARE_E_INVALID, capi.HARE_E_HIP)
    assert torch.cuda.current_device() == before


def test_bounce_c5_full_size_1M_rays_x8_1M_tris():
    """BASELINE config[4] at its per-GPU size: 1 048 576 rays x 8 specular bounces in the 1M-triangle cathedral, D = 128,
    device-resident; the events of EVERY bounce equal the oracle's, retired rays come back as X_Event()."""
    import torch
    m = H.scenes.cathedral()
    T, To = H.Topology(m.verts, m.nverts), po.Topology(m.verts, m.nverts)
    n, bounces = 1 << 20, 8
    rays = H.scenes.burst_rays(8 << 20, m.size, start=3 << 20, count=n)     # one rank's shard of the 8M-ray burst
    g = H.Voxel_Grid([T], 128)
    o = po.VoxelGrid([To], domain=128)
    d_rays = torch.from_numpy(rays.copy()).cuda()
    d_ev = torch.empty(n * 56, dtype=torch.uint8, device="cuda")
    d_ex = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    d_ctr = torch.zeros(8, dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    cur, excl = rays.copy(), np.full(n, -1, np.int32)
    dead = np.zeros(n, bool)
    casts = hits = 0
    for b in range(bounces):
        g.shoot_device(n, d_rays.data_ptr(), d_ev.data_ptr(), d_excl1=d_ex.data_ptr(), d_counters=d_ctr.data_ptr(), stream=st,
                       flags=capi.SHOOT_RETIRED_RAYS)
        g.reflect_device(n, d_rays.data_ptr(), d_ev.data_ptr(), d_ex.data_ptr(), stream=st)
        torch.cuda.synchronize()
        ev = np.frombuffer(d_ev.cpu().numpy().tobytes(), dtype=capi.XEVENT_DTYPE)
        live = ~dead
        ref = np.zeros(n, po.XEVENT_DTYPE)
        ref["poly_id"] = -1
        ref[live], c = o.shoot(cur[live], excl1=excl[live], nthreads=16)
        casts += c["rays"]
        hits += c["hits"]
        assert_events_equal(ev, ref, what=f"C5 full size, bounce {b}")
        alive = (ref["hit"] == 1) & live
        cur = po.reflect_batch(To, cur, ref)
        excl = np.where(alive, ref["poly_id"], -2).astype(np.int32)
        dead |= ~alive
        assert np.array_equal(d_rays.cpu().numpy(), cur)
        assert np.array_equal(d_ex.cpu().numpy(), excl)
    assert (int(d_ctr[0]), int(d_ctr[1])) == (casts, hits)
    assert dead.mean() < 0.01


# ---------------------------------------------------------------------------------------------------------------
# The voxel path has two production kernels (K1p hare_voxel_persist_*, K1q hare_voxel_pool_*); the library picks by
# batch size.  HARE_VOXEL_KERNEL forces one, so every case below runs on BOTH regardless of where the crossover sits.
@pytest.mark.parametrize("kernel", ["pool", "persist"])
def test_both_voxel_kernels_full_size_and_large_batches(hall, kernel, monkeypatch):
    monkeypatch.setenv("HARE_DEV", "1")     # developer overrides are read (once, at scene creation) only in a process that opted in
    monkeypatch.setenv("HARE_VOXEL_KERNEL", kernel)
    m, T, To = hall
    g = H.Voxel_Grid([T], 64)
    o = po.VoxelGrid([To], domain=64)
    for n in (1 << 20, 3_000_001):
        assert g.kernel_name(n) == f"hare_voxel_{kernel}_tri"
        rays = H.scenes.burst_rays(n, m.size)
        ev, c = g.Shoot_batch(rays)
        ref, rc = o.shoot(rays, nthreads=16)
        assert_events_equal(ev, ref, what=f"{kernel} kernel, {n} rays")
        assert (c["rays"], c["hits"]) == (n, rc["hits"])


@pytest.mark.parametrize("kernel", ["pool", "persist"])
def test_both_voxel_kernels_quads_exclusions_outside_origins_and_writeback(kernel, monkeypatch):
    monkeypatch.setenv("HARE_DEV", "1")     # developer overrides are read (once, at scene creation) only in a process that opted in
    monkeypatch.setenv("HARE_VOXEL_KERNEL", kernel)
    v, nv, size = soup()
    rays = soup_rays(30000, size)
    rng = np.random.default_rng(1)
    e1 = rng.integers(-3, len(nv), len(rays)).astype(np.int32)
    e2 = rng.integers(-1, len(nv), len(rays)).astype(np.int32)
    g = H.Voxel_Grid([H.Topology(v, nv)], 12)
    assert g.kernel_name(len(rays)) == f"hare_voxel_{kernel}_quad"
    o = po.VoxelGrid([po.Topology(v, nv)], domain=12)
    r1 = rays.copy()
    ev, _ = g.Shoot_batch(r1, poly_origin1=e1, poly_origin2=e2, writeback_origin=True)
    ref, _, moved = o.shoot(rays, excl1=e1, excl2=e2, mutate=True)
    assert_events_equal(ev, ref, what=f"{kernel}: soup with exclusions")
    assert r1.tobytes() == moved.tobytes() and (moved != rays).any()          # AABB.Intersect's origin move, written back
    ev, _ = g.Shoot_batch(rays)                                                # and without write-back: same events, rays untouched
    assert_events_equal(ev, o.shoot(rays)[0], what=f"{kernel}: soup, no write-back")


@pytest.mark.parametrize("kernel", ["pool", "persist"])
@pytest.mark.parametrize("domain", [1, 7, 33, 72, 96, 128, 200])
def test_both_voxel_kernels_over_grid_sizes(hall, kernel, domain, monkeypatch):
    """1 bit per voxel up to 80^3, per 2^3 block up to 160^3, per 4^3 above; K1q needs the bitmap to leave room for its
    pools (<= 32 KB), so 65..80 stay with K1p whatever is asked for -- the name tells."""
    monkeypatch.setenv("HARE_DEV", "1")     # developer overrides are read (once, at scene creation) only in a process that opted in
    monkeypatch.setenv("HARE_VOXEL_KERNEL", kernel)
    m, T, To = hall
    n = 150_000
    rays = H.scenes.burst_rays(n, m.size)
    g = H.Voxel_Grid([T], domain)
    name = g.kernel_name(n)
    assert name.startswith("hare_voxel_pool" if (kernel == "pool" and not 64 < domain <= 80) else "hare_voxel_persist"), name
    assert name.endswith("_g") == (domain > 80)
    ref, _ = po.VoxelGrid([To], domain=domain).shoot(rays, nthreads=16)
    assert_events_equal(g.Shoot_batch(rays)[0], ref, what=f"{kernel} D={domain}")


@pytest.mark.parametrize("kernel", ["pool", "persist"])
def test_both_voxel_kernels_degenerate_rays_and_two_topologies(kernel, monkeypatch):
    from tests.test_gpu_parity import bits_equal, degenerate_rays
    monkeypatch.setenv("HARE_DEV", "1")     # developer overrides are read (once, at scene creation) only in a process that opted in
    monkeypatch.setenv("HARE_VOXEL_KERNEL", kernel)
    m = H.scenes.shoebox()
    T, To = H.Topology(m.verts, m.nverts), po.Topology(m.verts, m.nverts)
    for domain in (1, 8, 33):
        rays = degenerate_rays(m.size, domain)
        ref, rc = po.VoxelGrid([To], domain=domain).shoot(rays.copy())
        ev, c = H.Voxel_Grid([T], domain).Shoot_batch(rays.copy())
        bits_equal(ev, ref, f"{kernel} D={domain} degenerate rays")
        assert c["hits"] == rc["hits"]
    v, nv, size = soup(150, 40)
    rays = soup_rays(5000, size)
    g = H.Voxel_Grid([T, H.Topology(v, nv)], 8)
    o = po.VoxelGrid([To, po.Topology(v, nv)], domain=8)
    for top in (0, 1):
        assert_events_equal(g.Shoot_batch(rays, top)[0], o.shoot(rays, top)[0], what=f"{kernel} top {top}")
    ev, c = g.Shoot_batch(rays[:0])
    assert len(ev) == 0 and c["rays"] == 0
    for n in (1, 63, 65, 129, 12289):                                          # ragged batches around the pool / wave sizes
        ev, c = g.Shoot_batch(rays[:n] if n <= len(rays) else np.resize(rays, (n, 6)))
        want = o.shoot(rays[:n] if n <= len(rays) else np.resize(rays, (n, 6)))[0]
        assert_events_equal(ev, want, what=f"{kernel} n={n}")
        assert c["rays"] == n


@pytest.mark.parametrize("kernel", ["pool", "persist"])
def test_both_voxel_kernels_bounce_loop_with_retired_rays(hall, kernel, monkeypatch):
    import torch
    monkeypatch.setenv("HARE_DEV", "1")     # developer overrides are read (once, at scene creation) only in a process that opted in
    monkeypatch.setenv("HARE_VOXEL_KERNEL", kernel)
    m, T, To = hall
    n, bounces = 200_000, 5
    rays = H.scenes.burst_rays(n, m.size)
    rays[::97, 3:] = rays[::97, 3:] * 0 + [0.0, 0.0, 0.0]                       # zero directions: miss at once, retired on bounce 1
    g = H.Voxel_Grid([T], 64)
    o = po.VoxelGrid([To], domain=64)
    d_rays = torch.from_numpy(rays.copy()).cuda()
    d_ev = torch.empty(n * 56, dtype=torch.uint8, device="cuda")
    d_ex = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    d_ctr = torch.zeros(8, dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    cur, excl = rays.copy(), np.full(n, -1, np.int32)
    dead = np.zeros(n, bool)
    casts = 0
    for b in range(bounces):
        g.shoot_device(n, d_rays.data_ptr(), d_ev.data_ptr(), d_excl1=d_ex.data_ptr(), d_counters=d_ctr.data_ptr(), stream=st,
                       flags=capi.SHOOT_RETIRED_RAYS)
        g.reflect_device(n, d_rays.data_ptr(), d_ev.data_ptr(), d_ex.data_ptr(), stream=st)
        torch.cuda.synchronize()
        ev = np.frombuffer(d_ev.cpu().numpy().tobytes(), dtype=capi.XEVENT_DTYPE)
        live = ~dead
        ref = np.zeros(n, po.XEVENT_DTYPE)
        ref["poly_id"] = -1
        ref[live], c = o.shoot(cur[live], excl1=excl[live], nthreads=16)
        casts += c["rays"]
        assert_events_equal(ev, ref, what=f"{kernel} bounce {b}")
        alive = (ref["hit"] == 1) & live
        cur = po.reflect_batch(To, cur, ref)
        excl = np.where(alive, ref["poly_id"], -2).astype(np.int32)
        dead |= ~alive
    assert int(d_ctr[0]) == casts and dead.sum() >= n // 97


@pytest.mark.parametrize("kernel", ["group", "dense", "pool", "persist"])
def test_both_octree_kernels(hall, kernel, monkeypatch):
    """K2g (hare_octree_group, the default: eight lanes per ray), K2p (hare_octree_persist, one lane per ray) and K2q (hare_octree_pool, opt-in: rays outnumber lanes, frames below the top
    one in a device scratch block): bench workload at 300k rays, tree shapes from a single leaf to 12 levels with
    quadrilaterals and exclusions, a 20-level tree, degenerate rays, and more launches in flight than K2q has scratch blocks."""
    import torch
    from tests.test_gpu_parity import bits_equal, degenerate_rays
    monkeypatch.setenv("HARE_DEV", "1")     # developer overrides are read (once, at scene creation) only in a process that opted in
    monkeypatch.setenv("HARE_OCTREE_KERNEL", kernel)
    m, T, To = hall
    n = 300_000
    rays = H.scenes.burst_rays(n, m.size)
    g = H.Octree([T], 8, 16)
    assert g.kernel_name(n) == f"hare_octree_{kernel}"
    ref, rc = po.Octree([To], 8, 16).shoot(rays, nthreads=16)
    ev, c = g.Shoot_batch(rays)
    assert_events_equal(ev, ref, what=f"octree {kernel}: hall")
    assert (c["rays"], c["hits"]) == (n, rc["hits"])
    v, nv, size = soup()
    sr = soup_rays(20000, size)
    rng = np.random.default_rng(1)
    e1 = rng.integers(-1, len(nv), len(sr)).astype(np.int32)
    e2 = rng.integers(-1, len(nv), len(sr)).astype(np.int32)
    for depth, maxp in ((0, 4), (1, 1), (3, 2), (6, 8), (12, 64)):
        gs, os_ = H.Octree([H.Topology(v, nv)], depth, maxp), po.Octree([po.Topology(v, nv)], depth, maxp)
        assert_events_equal(gs.Shoot_batch(sr)[0], os_.shoot(sr)[0], what=f"octree {kernel} depth {depth}")
        assert_events_equal(gs.Shoot_batch(sr, poly_origin1=e1, poly_origin2=e2)[0], os_.shoot(sr, excl1=e1, excl2=e2)[0],
                            what=f"octree {kernel} depth {depth} excl")
    dv, dnv, dr = deep_scene(20)
    assert_events_equal(H.Octree([H.Topology(dv, dnv)], 20, 2).Shoot_batch(dr)[0], po.Octree([po.Topology(dv, dnv)], 20, 2).shoot(dr)[0],
                        what=f"octree {kernel} 20 levels")
    sm = H.scenes.shoebox()
    deg = degenerate_rays(sm.size, 8)
    oref, _ = po.Octree([po.Topology(sm.verts, sm.nverts)], 5, 8).shoot(deg.copy())
    bits_equal(H.Octree([H.Topology(sm.verts, sm.nverts)], 5, 8).Shoot_batch(deg.copy())[0], oref, f"octree {kernel} degenerate rays")
    # eight launches back to back on one stream and on two streams: K2q owns 4 scratch blocks and orders re-use with events
    d_rays = torch.from_numpy(rays).cuda()
    outs = [torch.empty(n * 56, dtype=torch.uint8, device="cuda") for _ in range(8)]
    s2 = torch.cuda.Stream()
    for k in range(8):
        stream = torch.cuda.current_stream() if k % 2 == 0 else s2
        g.shoot_device(n, d_rays.data_ptr(), outs[k].data_ptr(), stream=stream.cuda_stream)
    torch.cuda.synchronize()
    for k in range(8):
        assert outs[k].cpu().numpy().tobytes() == ref.tobytes(), f"launch {k}"


@pytest.mark.parametrize("what", ["voxel", "octree_pool", "octree_persist", "octree_group", "octree_dense"])
def test_concurrent_batch_callers_on_one_scene(hall, what):
    """Pachyderm shoots from many worker threads at once: six host threads call Shoot_batch on ONE scene at the same time
    (four staging contexts: two of them wait their turn; up to 12 chunk streams launch side by side), different ray sets and
    sizes; every result equals the oracle's and the counters add up.  The octree pool kernel (K2q) keeps per-launch scratch
    blocks in a ring of four guarded by events: wait + launch + record of a block must stay together under its lock."""
    import threading
    m, T, To = hall
    if what == "voxel":
        g, o = H.Voxel_Grid([T], 64), po.VoxelGrid([To], domain=64)
        sizes = [300_000, 70_000, 1_000, 250_000, 33, 120_000]
    else:
        g, o = H.Octree([T], 8, 16), po.Octree([To], 8, 16)
        g.set_option("octree_kernel", {"octree_pool": 2, "octree_persist": 1, "octree_group": 3, "octree_dense": 4}[what])
        sizes = [200_000, 70_000, 1_000, 66_000, 33, 100_000]          # three chunks from 196 608 rays: several launches per call
        assert g.kernel_name(sizes[0]) == "hare_" + what
    burst = H.scenes.burst_rays(1 << 22, m.size)
    sets = [burst[k::7][:n].copy() for k, n in enumerate(sizes)]
    assert [len(r) for r in sets] == sizes
    refs = [o.shoot(r, nthreads=8) for r in sets]
    got = [None] * len(sets)
    errs = []

    def work(k):
        try:
            for _ in range(3):
                got[k] = g.Shoot_batch(sets[k])
        except Exception as e:      # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(k,)) for k in range(len(sets))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for k, ((ev, c), (ref, rc)) in enumerate(zip(got, refs)):
        assert_events_equal(ev, ref, what=f"thread {k}")
        assert (c["rays"], c["hits"]) == (sizes[k], rc["hits"])


def test_the_picker_follows_its_rule(hall):
    """launch.cpp choose_kernel: K1q for every batch size since round 3 (a batch below one pool fill of the chip is spread over all
    waves, which the wide drain modes then serve with several lanes per ray); K1p for grids the pool kernel cannot take (more than
    512 voxels a side); the per-scene option is for A/B runs."""
    _, T, _ = hall
    g = H.Voxel_Grid([T], 64)
    for n in (1, 64, 65536, 393215, 393216, 1 << 20, 1 << 24):
        assert g.kernel_name(n) == "hare_voxel_pool_tri", (n, g.kernel_name(n))
    g128 = H.Voxel_Grid([T], 128)                        # a coarse occupancy bitmap (one bit per 2^3 voxels): the same rule
    assert g128.kernel_name(1000) == "hare_voxel_pool_tri_g" and g128.kernel_name(1 << 20) == "hare_voxel_pool_tri_g"
    g.set_option("voxel_kernel", 2)                      # the per-scene switch the A/B tests and tools use
    assert g.kernel_name(64) == "hare_voxel_pool_tri"
    g.set_option("voxel_kernel", 1)
    assert g.kernel_name(1 << 24) == "hare_voxel_persist_tri"
    with pytest.raises(H.HareError):
        g.set_option("no_such_option", 1)
