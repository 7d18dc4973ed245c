"""Round 6.  K1q's DDA step loop written by hand for gfx950 (hare_amd/csrc/voxel_walk.h; scene option `voxel_walk`): the per-axis updates
of Voxel_Grid.cs:713-759 under the axis' own EXEC mask instead of selects, in the pool's ordinary walk and in the wide walk of the drain.
It must be the same steps in the same order -- same voxels, same tMax bit patterns, same events -- so every case is run three ways: the
hand-written loop, the compiler's loop, the oracle.  The cases sit where a step loop can go wrong: grids of 1 ... 200 voxels a side (a
bit per voxel, per 2^3 block at 128, per 4^3 at 200), origins outside the grid and exactly on voxel faces, directions with zero /
negative-zero / denormal / huge components and NaN (every compare false: the z axis steps), both exclusions, exact ties, quadrilaterals,
batches of one ray ... a million (a thin batch runs the wide walk from its second round on), the bounce loop cast by cast."""
import numpy as np
import pytest

import hare_amd as H
from oracle import pyoracle as po
from tests.helpers import assert_events_equal, oracle_bounce_loop, soup, soup_rays
from tests.test_gpu_ties import tie_rays, tie_scene

pytestmark = pytest.mark.gpu


def three_ways(g, o, rays, what, **kw):
    okw = {("excl1" if k == "poly_origin1" else "excl2"): v for k, v in kw.items() if k.startswith("poly_origin")}
    ref, rc = o.shoot(rays, nthreads=16, **okw)
    for walk in (1, 0):
        g.set_option("voxel_walk", walk)
        assert g.get_option("voxel_walk") == walk
        ev, c = g.Shoot_batch(rays, **kw)
        assert_events_equal(ev, ref, what=f"{what} voxel_walk={walk}")
        assert c["hits"] == rc["hits"]
    g.set_option("voxel_walk", 1)
    return ref


def awkward_rays(size, vd_hint, n=24_000, seed=5):
    """Origins inside, outside and exactly on multiples of a voxel edge; directions with exact zeros, negative zeros, denormals, huge and
    tiny magnitudes, axis-parallel, exact diagonals (tMax ties on every step), NaN and infinite components."""
    rng = np.random.default_rng(seed)
    r = soup_rays(n, size, seed=seed + 1)
    k = np.arange(n)
    r[k % 11 == 0, 3] = 0.0
    r[k % 11 == 1, 4] = -0.0
    r[k % 11 == 2, 5] = 5e-324
    r[k % 13 == 3, 3:] *= 1e200
    r[k % 13 == 4, 3:] *= 1e-200
    ax = k % 17 == 5
    r[ax, 3:] = np.eye(3)[rng.integers(0, 3, ax.sum())] * rng.choice([-1.0, 1.0], ax.sum())[:, None]
    dg = k % 17 == 6
    r[dg, 3:] = rng.choice([-1.0, 1.0], (dg.sum(), 3))                        # |dx| = |dy| = |dz|: the three tMax sequences tie again and again
    on = k % 7 == 3
    r[on, :3] = np.round(r[on, :3] / vd_hint) * vd_hint                       # origins on voxel faces / edges / corners of a lattice
    r[k % 97 == 9, 3] = np.nan
    r[k % 97 == 10, 1] = np.nan
    r[k % 97 == 11, 4] = np.inf
    r[k % 97 == 12, 5] = -np.inf
    return np.ascontiguousarray(r)


def test_hand_written_walk_equals_the_compilers_and_the_oracle_on_awkward_rays():
    v, nv, size = soup(n_tri=700, n_quad=300, seed=21)
    T, To = H.Topology(v, nv), po.Topology(v, nv)
    rng = np.random.default_rng(1)
    for D in (1, 2, 7, 8, 33, 64, 67, 128, 200):
        g, o = H.Voxel_Grid([T], D), po.VoxelGrid([To], domain=D)
        rays = awkward_rays(size, vd_hint=(size[0] + 0.202) / D, seed=D)
        assert g.kernel_name(len(rays)).startswith("hare_voxel_pool"), (D, g.kernel_name(len(rays)))
        ref = three_ways(g, o, rays, f"awkward D={D}")
        e1 = np.where(rng.random(len(rays)) < 0.5, ref["poly_id"], rng.integers(-1, len(nv), len(rays))).astype(np.int32)
        e2 = rng.integers(-1, len(nv), len(rays)).astype(np.int32)
        three_ways(g, o, rays, f"awkward D={D} excl", poly_origin1=e1, poly_origin2=e2)
        if D in (8, 64, 128):                       # origin write-back (Voxel_Grid.cs:573-581 moves R): the moved origins too
            refm, _, moved = o.shoot(rays, mutate=True)
            for walk in (1, 0):
                g.set_option("voxel_walk", walk)
                r = rays.copy()
                ev, _ = g.Shoot_batch(r, writeback_origin=True)
                ok = ~np.isnan(moved).any(axis=1)   # NaN payload bits of a moved NaN origin are not part of the contract
                assert_events_equal(ev, refm, what=f"awkward D={D} write-back voxel_walk={walk}")
                assert np.array_equal(r[ok].view(np.int64), moved[ok].view(np.int64))
                assert np.array_equal(np.isnan(r), np.isnan(moved))
            g.set_option("voxel_walk", 1)


def test_hand_written_walk_on_exact_ties():
    v, nv, size = tie_scene()
    T, To = H.Topology(v, nv), po.Topology(v, nv)
    rays = tie_rays(v, nv, size, n=6000)
    for D in (1, 8, 21, 64, 128):
        g, o = H.Voxel_Grid([T], D), po.VoxelGrid([To], domain=D)
        ref = three_ways(g, o, rays, f"ties D={D}")
        three_ways(g, o, rays, f"ties D={D} excl", poly_origin1=ref["poly_id"].astype(np.int32))


@pytest.mark.parametrize("scene,domain,n", [("hall", 64, 1 << 20), ("hall", 128, 300_000), ("hall", 200, 300_000), ("cathedral", 128, 400_000)])
def test_hand_written_walk_at_bench_scale(scene, domain, n):
    """The bench's own workloads: the 1M-ray burst into the hall at D = 64 in one piece; the per-block bitmaps (D = 128: 2^3, D = 200: 4^3)."""
    m = H.scenes.SCENES[scene]()
    T, To = H.Topology(m.verts, m.nverts), po.Topology(m.verts, m.nverts)
    g, o = H.Voxel_Grid([T], domain), po.VoxelGrid([To], domain=domain)
    rays = H.scenes.burst_rays(n, m.size)
    ref, rc = o.shoot(rays, nthreads=32)
    for walk in (1, 0):
        g.set_option("voxel_walk", walk)
        for wide in (1, 0):                         # the wide walk of the drain runs the hand-written loop too
            g.set_option("wide_drain", wide)
            ev, c = g.Shoot_batch(rays)
            assert_events_equal(ev, ref, what=f"{scene} D={domain} voxel_walk={walk} wide_drain={wide}")
            assert c["hits"] == rc["hits"]
    g.set_option("voxel_walk", 1); g.set_option("wide_drain", 1)
    # thin batches: spread over all waves, in the drain (wide walk, cooperative tail) from the second round on
    for k in (1, 63, 64, 1000, 9216, 70_000):
        for walk in (1, 0):
            g.set_option("voxel_walk", walk)
            ev, _ = g.Shoot_batch(rays[:k])
            assert_events_equal(ev, ref[:k], what=f"{scene} D={domain} {k} rays voxel_walk={walk}")
    g.set_option("voxel_walk", 1)


def test_hand_written_walk_in_the_bounce_loop():
    """Reflected rays start on a polygon, often exactly on a voxel face, and skim walls: cast by cast against the oracle's loop, a launch
    per cast and the one-launch loop, hand-written and compiler's step loop."""
    for scene, D, n in (("hall", 64, 120_000), ("cathedral", 128, 60_000)):
        m = H.scenes.SCENES[scene]()
        T, To = H.Topology(m.verts, m.nverts), po.Topology(m.verts, m.nverts)
        rays = H.scenes.burst_rays(n, m.size)
        g, o = H.Voxel_Grid([T], D), po.VoxelGrid([To], domain=D)
        ref, rc = oracle_bounce_loop(po, To, o, rays, 6)
        for walk in (1, 0):
            g.set_option("voxel_walk", walk)
            for fused in (0, 1):
                g.set_option("bounce_fused", fused)
                ev, c, pcs = g.Bounce_batch(rays, 6, per_cast=True, all_casts=True)
                for b in range(6):
                    assert_events_equal(ev[b], ref[b], what=f"{scene} bounce cast {b} voxel_walk={walk} fused={fused}")
                assert [(p["rays"], p["hits"]) for p in pcs] == [(p["rays"], p["hits"]) for p in rc]
        g.set_option("bounce_fused", 0); g.set_option("voxel_walk", 1)


def test_own_work_counters_do_not_depend_on_the_step_loop():
    """HARE_SHOOT_COUNT_OWN counts the voxels the kernel walks into inside the hand-written loop (a per-lane counter under EXEC): the same
    totals as the compiler's loop counts (to within the drain's re-scans), ray for ray the same events."""
    import torch
    from hare_amd import capi
    for scene, D in (("hall", 64), ("cathedral", 128)):
        m = H.scenes.SCENES[scene]()
        g = H.Voxel_Grid([H.Topology(m.verts, m.nverts)], D)
        n = 300_000
        d_rays = torch.from_numpy(H.scenes.burst_rays(n, m.size)).cuda()
        d_out = torch.empty(n * 56, dtype=torch.uint8, device="cuda")
        got = {}
        for walk in (1, 0):
            g.set_option("voxel_walk", walk)
            d_ctr = torch.zeros(8, dtype=torch.int64, device="cuda")
            assert g.kernel_name(n, flags=capi.SHOOT_COUNT_OWN).endswith("_own")
            g.shoot_device(n, d_rays.data_ptr(), d_out.data_ptr(), d_counters=d_ctr.data_ptr(), flags=capi.SHOOT_COUNT_OWN)
            torch.cuda.synchronize()
            got[walk] = (d_out.cpu().numpy().tobytes(), [int(x) for x in d_ctr.cpu()])
        g.set_option("voxel_walk", 1)
        assert got[1][0] == got[0][0], scene
        # the same rays and hits; the work counters agree to within what the two loops' different task boundaries make of the drain (a ray the
        # cooperative tail or a wide mode picks up re-scans its voxel's list): well under 2 %
        a, b = got[1][1], got[0][1]
        assert a[:2] == b[:2] == [n, a[1]], (scene, a, b)
        for k in (2, 3, 4, 5):
            assert abs(a[k] - b[k]) <= 0.02 * b[k], (scene, a, b)
        assert a[2] > 10 * n                                        # voxels walked into


def test_a_shoot_never_allocates_frees_or_waits():
    """VERDICT round 5 / ADVICE: hare_shoot_device is documented as stream-ordered, so the FIRST large batch on a fresh grid -- the one that
    used to grow the pool kernel's order ring inside the call (hipFree = a device-wide synchronisation) -- must make no hipMalloc, no
    hipFree and no host-side wait.  The HIP runtime is bound through hiprt.cpp, which counts those calls (hare_scene_get_option).  The
    ring is reserved with the grid: "voxel_order_bytes" says so; a batch beyond "voxel_order_max_rays", or a ring switched off, runs in
    the caller's order with the same events.  The octree's first launches likewise (scratch reserved with the tree, and only what the
    scene's options can launch: nothing for the library's own kernel rule on a shallow tree)."""
    import torch
    from hare_amd import capi
    m = H.scenes.hall()
    T = H.Topology(m.verts, m.nverts)
    g = H.Voxel_Grid([T], 64)
    assert g.get_option("voxel_order_max_rays") == 1 << 24
    assert g.get_option("voxel_order_bytes") == 4 * (1 << 24) * 4
    n = 2 << 20
    d_rays = torch.from_numpy(H.scenes.burst_rays(n, m.size)).cuda()
    d_out = torch.empty(n * 56, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    before = [g.get_option(k) for k in ("hip_malloc_calls", "hip_free_calls", "hip_sync_calls")]
    for _ in range(6):                          # more launches than the ring has blocks: the fifth waits on its stream, not on the host
        g.shoot_device(n, d_rays.data_ptr(), d_out.data_ptr())
    after = [g.get_option(k) for k in ("hip_malloc_calls", "hip_free_calls", "hip_sync_calls")]
    assert after == before, (before, after)
    torch.cuda.synchronize()
    ordered = d_out.cpu().numpy().tobytes()
    g.set_option("voxel_order_max_rays", 1 << 20)          # smaller than the batch: no order pass
    assert g.get_option("voxel_order_bytes") == 4 * (1 << 20) * 4
    g.shoot_device(n, d_rays.data_ptr(), d_out.data_ptr())
    torch.cuda.synchronize()
    assert d_out.cpu().numpy().tobytes() == ordered
    g.set_option("voxel_order", 0)
    assert g.get_option("voxel_order_bytes") == 0
    g.shoot_device(n, d_rays.data_ptr(), d_out.data_ptr())
    torch.cuda.synchronize()
    assert d_out.cpu().numpy().tobytes() == ordered
    g.set_option("voxel_order", 2); g.set_option("voxel_order_max_rays", 1 << 22)      # every batch ordered, exclusions too
    assert g.get_option("voxel_order_bytes") == 4 * (1 << 22) * 4
    before = [g.get_option(k) for k in ("hip_malloc_calls", "hip_free_calls", "hip_sync_calls")]
    g.shoot_device(n, d_rays.data_ptr(), d_out.data_ptr())
    assert [g.get_option(k) for k in ("hip_malloc_calls", "hip_free_calls", "hip_sync_calls")] == before
    torch.cuda.synchronize()
    assert d_out.cpu().numpy().tobytes() == ordered

    oc = H.Octree([T], 8, 16)
    small = oc.get_option("octree_scratch_bytes")           # the library's rule (K2g / K2d, no hand-over): K2g's stack spill alone
    assert 0 <= small <= 256 << 20
    before = [oc.get_option(k) for k in ("hip_malloc_calls", "hip_free_calls", "hip_sync_calls")]
    for k in (100_000, 1 << 20):
        oc.shoot_device(k, d_rays.data_ptr(), d_out.data_ptr())
    assert [oc.get_option(k) for k in ("hip_malloc_calls", "hip_free_calls", "hip_sync_calls")] == before
    torch.cuda.synchronize()
    want = d_out[: (1 << 20) * 56].cpu().numpy().tobytes()
    oc.set_option("octree_kernel", 1)                       # K2p hands rays over: its records are reserved now, not in the launch
    assert oc.get_option("octree_scratch_bytes") > 2 * small
    before = [oc.get_option(k) for k in ("hip_malloc_calls", "hip_free_calls", "hip_sync_calls")]
    oc.shoot_device(1 << 20, d_rays.data_ptr(), d_out.data_ptr())
    assert [oc.get_option(k) for k in ("hip_malloc_calls", "hip_free_calls", "hip_sync_calls")] == before
    torch.cuda.synchronize()
    assert d_out[: (1 << 20) * 56].cpu().numpy().tobytes() == want


def test_exact_block_skip_option_changes_nothing_but_the_step_count():
    """Scene option `voxel_skip`: K1q's walk crosses an EMPTY aligned block of 4^3 voxels in one operation --
    the closed-form construction (the DDA as a merge of three sequences of sequential adds).  It must land in the same voxel with the same tMax
    bit patterns, so: the same events with the option on, off and from the oracle -- at D = 64 (a bit per voxel), 128 (per 2^3 block), 200 (per
    4^3 block) and sizes that are no multiple of four, on awkward rays (origins outside the grid and on voxel faces, zero / denormal / huge / NaN
    components: lanes that are not finite step), exact diagonals and ties, both exclusions, the bench's burst, and the bounce loop cast by cast.
    The counting build says what it does: fewer operations executed (word 6) for the same voxels crossed (word 2)."""
    import torch
    from hare_amd import capi
    v, nv, size = soup(n_tri=700, n_quad=300, seed=21)
    T, To = H.Topology(v, nv), po.Topology(v, nv)
    rng = np.random.default_rng(2)
    for D in (7, 33, 64, 66, 128, 200):
        g, o = H.Voxel_Grid([T], D), po.VoxelGrid([To], domain=D)
        rays = awkward_rays(size, vd_hint=(size[0] + 0.202) / D, seed=100 + D)
        ref, rc = o.shoot(rays, nthreads=16)
        e1 = np.where(rng.random(len(rays)) < 0.5, ref["poly_id"], rng.integers(-1, len(nv), len(rays))).astype(np.int32)
        e2 = rng.integers(-1, len(nv), len(rays)).astype(np.int32)
        refx, _ = o.shoot(rays, excl1=e1, excl2=e2, nthreads=16)
        for skip in (1, 0):
            g.set_option("voxel_skip", skip)
            assert g.get_option("voxel_skip") == skip
            assert_events_equal(g.Shoot_batch(rays)[0], ref, what=f"awkward D={D} voxel_skip={skip}")
            assert_events_equal(g.Shoot_batch(rays, poly_origin1=e1, poly_origin2=e2)[0], refx, what=f"awkward D={D} excl voxel_skip={skip}")
    v, nv, size = tie_scene()
    T, To = H.Topology(v, nv), po.Topology(v, nv)
    rays = tie_rays(v, nv, size, n=6000)
    for D in (8, 21, 64, 128):
        g, o = H.Voxel_Grid([T], D), po.VoxelGrid([To], domain=D)
        g.set_option("voxel_skip", 1)
        assert_events_equal(g.Shoot_batch(rays)[0], o.shoot(rays)[0], what=f"ties D={D} voxel_skip=1")
    for scene, D, n in (("hall", 64, 400_000), ("hall", 128, 200_000), ("hall", 200, 200_000), ("cathedral", 128, 300_000)):
        m = H.scenes.SCENES[scene]()
        T, To = H.Topology(m.verts, m.nverts), po.Topology(m.verts, m.nverts)
        g, o = H.Voxel_Grid([T], D), po.VoxelGrid([To], domain=D)
        rays = H.scenes.burst_rays(n, m.size)
        ref, rc = o.shoot(rays, nthreads=32)
        got = {}
        d_rays = torch.from_numpy(rays).cuda()
        d_out = torch.empty(n * 56, dtype=torch.uint8, device="cuda")
        for skip in (1, 0):
            g.set_option("voxel_skip", skip)
            ev, c = g.Shoot_batch(rays)
            assert_events_equal(ev, ref, what=f"{scene} D={D} voxel_skip={skip}")
            assert c["hits"] == rc["hits"]
            for k in (1, 64, 5000):
                assert_events_equal(g.Shoot_batch(rays[:k])[0], ref[:k], what=f"{scene} D={D} {k} rays voxel_skip={skip}")
            d_ctr = torch.zeros(8, dtype=torch.int64, device="cuda")
            g.shoot_device(n, d_rays.data_ptr(), d_out.data_ptr(), d_counters=d_ctr.data_ptr(), flags=capi.SHOOT_COUNT_OWN)
            torch.cuda.synchronize()
            got[skip] = [int(x) for x in d_ctr.cpu()]
        # the same voxels crossed (to within the drain's re-walks), in fewer operations
        assert abs(got[1][2] - got[0][2]) <= 0.02 * got[0][2], (scene, D, got)
        assert got[1][6] < 0.8 * got[1][2] and got[0][6] >= 0.95 * got[0][2], (scene, D, got)
        if scene == "hall" and D == 64:
            refb, rcb = oracle_bounce_loop(po, To, o, rays[:100_000], 5)
            for skip in (1, 0):
                g.set_option("voxel_skip", skip)
                evb, _, pcs = g.Bounce_batch(rays[:100_000], 5, per_cast=True, all_casts=True)
                for b in range(5):
                    assert_events_equal(evb[b], refb[b], what=f"bounce cast {b} voxel_skip={skip}")
        g.set_option("voxel_skip", 0)


def test_trees_do_not_test_the_polygon_of_their_hit_again():
    """`Octree.Shoot` walks on behind a hit (F15) and a polygon is listed in every leaf it touches: three of four exact tests of K2d were the
    SAME ray against the SAME polygon its hit lies on -- the same t bit for bit, never `< closestT`, nothing changes.  The tree kernels skip
    those (HARE_K2D_SKIP_PID: K2d, K2g, K2p, K3d).  Events equal to the oracle's (which tests them all); the counting builds show the repeats
    gone: at most 1.5 exact tests per ray where the oracle makes 4 and more."""
    import torch
    from hare_amd import capi
    m = H.scenes.hall()
    T, To = H.Topology(m.verts, m.nverts), po.Topology(m.verts, m.nverts)
    n = 200_000
    rays = H.scenes.burst_rays(1 << 20, m.size)[::5][:n].copy()
    d_rays = torch.from_numpy(rays).cuda()
    d_out = torch.empty(n * 56, dtype=torch.uint8, device="cuda")
    for name, g, o in (("octree", H.Octree([T], 8, 16), po.Octree([To], 8, 16)), ("kdtree", H.KDTree([T], 16, 8), po.KDTree([To], 16, 8))):
        ref, rc = o.shoot(rays[:20_000], nthreads=16)
        if name == "octree":
            g.set_option("octree_kernel", 4)
        d_ctr = torch.zeros(8, dtype=torch.int64, device="cuda")
        assert g.kernel_name(n, flags=capi.SHOOT_COUNT_OWN).endswith("_dense_own")
        g.shoot_device(n, d_rays.data_ptr(), d_out.data_ptr(), d_counters=d_ctr.data_ptr(), flags=capi.SHOOT_COUNT_OWN)
        torch.cuda.synchronize()
        ev = np.frombuffer(d_out.cpu().numpy().tobytes(), dtype=ref.dtype)
        assert_events_equal(ev[:20_000], ref, what=name)
        c = [int(x) for x in d_ctr.cpu()]
        assert c[0] == n and c[1] > 0.9 * n, c
        assert c[4] <= 1.5 * n, (name, "exact tests per ray", c[4] / n)
        assert rc["tests"] >= 3.0 * len(ref) or name == "kdtree", (name, rc)       # what the reference's walk makes of the same rays


def test_host_buffer_calls_write_into_the_callers_array_and_only_read_its_rays():
    """The Python mirror's host-buffer calls with `out=`: the events land in the caller's array (the same object comes back), byte for byte
    what a call without it returns and what the oracle says, call after call into the same array (a stale record would show: the second
    batch differs from the first), sharded or not, the bounce loop too.  And a call without `writeback_origin` only READS the rays: the
    mirror passes the caller's array itself (no defensive copy any more) and its bytes are what they were."""
    from hare_amd import capi
    m = H.scenes.hall()
    T, To = H.Topology(m.verts, m.nverts), po.Topology(m.verts, m.nverts)
    g, g2, o = H.Voxel_Grid([T], 64), H.Voxel_Grid([T], 64), po.VoxelGrid([To], domain=64)
    n = 50_000
    all_rays = H.scenes.burst_rays(1 << 20, m.size)
    ra, rb = all_rays[:n].copy(), all_rays[7 * n:8 * n].copy()
    ra[::3, :3] += 40.0                                      # origins outside the grid: the rays a write-back would move
    SP = H.Spatial_Partition
    out = np.zeros(n, capi.XEVENT_DTYPE)
    for rays in (ra, rb):
        before = rays.tobytes()
        ref, rc = o.shoot(rays, nthreads=16)
        ev, c = g.Shoot_batch(rays, out=out)
        assert ev is out and c["hits"] == rc["hits"]
        assert_events_equal(out, ref, what="Shoot_batch out=")
        assert out.tobytes() == g.Shoot_batch(rays)[0].tobytes()
        out[:] = 0
        ev, c = SP.Shoot_batch_sharded([g, g2], rays, out=out)
        assert ev is out and c["hits"] == rc["hits"]
        assert_events_equal(out, ref, what="Shoot_batch_sharded out=")
        assert rays.tobytes() == before
    sl = np.zeros(n, capi.SLIM_DTYPE)
    assert g.Shoot_batch(rb, slim=True, out=sl)[0] is sl
    assert g.expand_events(rb, sl).tobytes() == out.tobytes()
    refb, _ = oracle_bounce_loop(po, To, o, rb, 4)
    before = rb.tobytes()
    allc = np.zeros((4, n), capi.XEVENT_DTYPE)
    assert g.Bounce_batch(rb, 4, all_casts=True, out=allc)[0] is allc
    assert SP.Bounce_batch_sharded([g, g2], rb, 4, out=out)[0] is out
    for b in range(4):
        assert_events_equal(allc[b], refb[b], what=f"Bounce_batch out=, cast {b}")
    assert_events_equal(out, refb[3], what="Bounce_batch_sharded out=, last cast")
    assert rb.tobytes() == before
