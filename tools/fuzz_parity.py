"""Developer tool: a long randomized differential run of the HIP paths against the oracle -- many more seeds than the test
suite carries, random grid sizes (incl. the coarse-bitmap range and the 65..80 range the pool kernel cannot take), random tree
shapes, random batch sizes (ragged waves), both voxel kernels and both octree kernels forced in turn, exclusions, origin
write-back.  Stops at the first difference and prints the configuration that reproduces it.

    SEEDS=0:300 python tools/fuzz_parity.py
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hare_amd as H
from oracle import pyoracle as po
from tests.test_gpu_parity import _random_scene

F = ("hit", "poly_id", "t", "u", "v", "x", "y", "z")


def same(a, b):
    for f in F:
        x, y = np.ascontiguousarray(a[f]), np.ascontiguousarray(b[f])
        if x.dtype.kind == "f": x, y = x.view(np.int64), y.view(np.int64)
        bad = np.nonzero(x != y)[0]
        if bad.size: return "X_Event.%s differs on rays %s" % (f, bad[:6])
    return None


def main():
    lo, hi = (int(x) for x in os.environ.get("SEEDS", "0:200").split(":"))
    t0 = time.time(); checks = 0
    for seed in range(lo, hi):
        verts, nverts, rays = _random_scene(seed)
        rng = np.random.default_rng(10_000 + seed)
        n = int(rng.choice([1, 63, 64, 65, 1000, 4000]))
        rays = np.ascontiguousarray(rays[:n])
        T, To = H.Topology(verts, nverts), po.Topology(verts, nverts)
        D = int(rng.choice([1, 2, 5, 13, 31, 64, 70, 80, 81, 97, 128, 161, 200]))
        e1 = rng.integers(-1, len(nverts), n).astype(np.int32); e2 = rng.integers(-1, len(nverts), n).astype(np.int32)
        g, o = H.Voxel_Grid([T], D), po.VoxelGrid([To], domain=D)
        ref, _ = o.shoot(rays); refx, _ = o.shoot(rays, excl1=e1, excl2=e2)
        moved = rays.copy(); refm, _, movedref = o.shoot(rays, mutate=True)
        for kern in ("persist", "pool"):
            g.set_option("voxel_kernel", {"persist": 1, "pool": 2}[kern])
            cfg = "seed %d voxel D=%d n=%d kernel=%s (%s)" % (seed, D, n, kern, g.kernel_name(n))
            for what, got, want in (("plain", g.Shoot_batch(rays)[0], ref), ("excl", g.Shoot_batch(rays, poly_origin1=e1, poly_origin2=e2)[0], refx)):
                bad = same(got, want); checks += 1
                if bad: print("MISMATCH", cfg, what, bad); return 1
            r = rays.copy(); got, _ = g.Shoot_batch(r, writeback_origin=True); checks += 1
            bad = same(got, refm) or (None if np.array_equal(r.view(np.int64), movedref.view(np.int64)) else "moved origins differ")
            if bad: print("MISMATCH", cfg, "writeback", bad); return 1
        g.set_option("voxel_kernel", 0)
        # round 6: the pool kernel's step loop as the compiler writes it (the default is the hand-written one, voxel_walk.h), the drain's wide modes off too
        g.set_option("voxel_kernel", 2); g.set_option("voxel_walk", 0)
        for wide in (1, 0):
            g.set_option("wide_drain", wide)
            cfg = "seed %d voxel D=%d n=%d pool kernel, compiler's step loop, wide_drain %d" % (seed, D, n, wide)
            for what, got, want in (("plain", g.Shoot_batch(rays)[0], ref), ("excl", g.Shoot_batch(rays, poly_origin1=e1, poly_origin2=e2)[0], refx)):
                bad = same(got, want); checks += 1
                if bad: print("MISMATCH", cfg, what, bad); return 1
        g.set_option("voxel_walk", 1)
        if seed % 3 == 1:       # ... and the hand-written loop with the wide drain off (the pool's ordinary walk to the end)
            g.set_option("wide_drain", 0)
            bad = same(g.Shoot_batch(rays)[0], ref); checks += 1
            if bad: print("MISMATCH seed %d voxel D=%d n=%d hand-written loop, wide_drain 0" % (seed, D, n), bad); return 1
        g.set_option("wide_drain", 1)
        if seed % 2 == 0:       # ... and the exact multi-voxel skip (scene option voxel_skip: empty 4^3 blocks crossed in one operation)
            g.set_option("voxel_skip", 1)
            for what, got, want in (("plain", g.Shoot_batch(rays)[0], ref), ("excl", g.Shoot_batch(rays, poly_origin1=e1, poly_origin2=e2)[0], refx)):
                bad = same(got, want); checks += 1
                if bad: print("MISMATCH seed %d voxel D=%d n=%d voxel_skip %s" % (seed, D, n, what), bad); return 1
            g.set_option("voxel_skip", 0)
        g.set_option("voxel_kernel", 0)
        if seed % 5 == 2:
            # round 6: hare_bounce_batch, last cast's events only: the launch-per-cast loop with the live-block list (bounce_pack 1: from
            # 4 096 rays) and without, against the oracle's loop -- open soups: rays leave, whole blocks of 64 die
            from tests.helpers import oracle_bounce_loop
            rb = np.ascontiguousarray(np.concatenate([rays, rays[::-1] * np.array([1, 1, 1, -1, -1, -1.0])]))[:int(rng.choice([4096, 5000, 8000]))]
            refb, _ = oracle_bounce_loop(po, To, o, rb, 5)
            for pack in (1, 0):
                g.set_option("bounce_pack", pack)
                evb, _ = g.Bounce_batch(rb, 5)
                bad = same(evb, refb[4]); checks += 1
                if bad: print("MISMATCH seed %d voxel D=%d bounce loop n=%d bounce_pack %d" % (seed, D, len(rb), pack), bad); return 1
            g.set_option("bounce_pack", 1)
        depth, maxp = int(rng.integers(0, 9)), int(rng.integers(1, 40))
        # the reference pads child boxes by an ABSOLUTE 0.1 m ("Octree - alt.cs":99-111): below ~0.4 m a node's
        # polygons land in all eight children and the tree grows 8x per level in ANY implementation -- keep nodes above 1 m
        ext = float((verts[:, :3].reshape(-1, 3).max(0) - verts[:, :3].reshape(-1, 3).min(0)).max())
        depth = min(depth, max(0, int(np.floor(np.log2(max(ext, 1e-9))))))
        oc, oo = H.Octree([T], depth, maxp), po.Octree([To], depth, maxp)
        oref, _ = oo.shoot(rays); orefx, _ = oo.shoot(rays, excl1=e1, excl2=e2)
        for kern in ("group", "dense", "persist", "pool"):
            oc.set_option("octree_kernel", {"persist": 1, "pool": 2, "group": 3, "dense": 4}[kern])
            cfg = "seed %d octree %d/%d n=%d kernel=%s" % (seed, depth, maxp, n, kern)
            for what, got, want in (("plain", oc.Shoot_batch(rays)[0], oref), ("excl", oc.Shoot_batch(rays, poly_origin1=e1, poly_origin2=e2)[0], orefx)):
                bad = same(got, want); checks += 1
                if bad: print("MISMATCH", cfg, what, bad); return 1
        oc.set_option("octree_kernel", 0)

        def occl_check(part, name, want_ev, r, ex1, ex2):
            """The flags-only kernels (hare_*_occl*: a ray ends as soon as its flag is decided) against the closest-hit predicate."""
            nonlocal checks
            hits = want_ev["t"][want_ev["hit"] == 1]
            scale = float(np.median(hits)) if hits.size else 1.0
            for tm in (None, np.abs(rng.normal(0, 1, r.shape[0])) * scale):
                occ = part.Occluded_batch(r, t_max=tm, poly_origin1=ex1, poly_origin2=ex2, events=False)[0]
                want = (want_ev["hit"] == 1) if tm is None else ((want_ev["hit"] == 1) & (want_ev["t"] < tm))
                checks += 1
                if not np.array_equal(np.asarray(occ, bool), want):
                    print("MISMATCH seed %d %s flags-only occlusion, t_max %s, %d flags differ" % (seed, name, "none" if tm is None else "drawn", int((np.asarray(occ, bool) != want).sum())))
                    return False
            return True
        if seed % 2 == 1:
            if not occl_check(oc, "octree %d/%d n=%d" % (depth, maxp, n), oref, rays, None, None): return 1
            if not occl_check(oc, "octree %d/%d n=%d excl" % (depth, maxp, n), orefx, rays, e1, e2): return 1
            if not occl_check(g, "voxel D=%d n=%d" % (D, n), ref, rays, None, None): return 1
            if not occl_check(g, "voxel D=%d n=%d excl" % (D, n), refx, rays, e1, e2): return 1
        if seed % 5 == 0 and n >= 1000:
            # the device-resident bounce loop on this scene (open soups: many rays leave and are retired), both voxel kernels,
            # and the occlusion predicate on the first cast
            import torch
            from hare_amd import capi
            st = torch.cuda.current_stream().cuda_stream
            for kern in ("persist", "pool"):
                g.set_option("voxel_kernel", {"persist": 1, "pool": 2}[kern])
                d_rays = torch.from_numpy(rays.copy()).cuda(); d_ev = torch.empty(n * 56, dtype=torch.uint8, device="cuda")
                d_ex = torch.full((n,), -1, dtype=torch.int32, device="cuda")
                cur, excl, dead = rays.copy(), np.full(n, -1, np.int32), np.zeros(n, bool)
                for b in range(4):
                    g.shoot_device(n, d_rays.data_ptr(), d_ev.data_ptr(), d_excl1=d_ex.data_ptr(), stream=st, flags=capi.SHOOT_RETIRED_RAYS)
                    g.reflect_device(n, d_rays.data_ptr(), d_ev.data_ptr(), d_ex.data_ptr(), stream=st)
                    torch.cuda.synchronize()
                    ev = np.frombuffer(d_ev.cpu().numpy().tobytes(), dtype=capi.XEVENT_DTYPE)
                    live = ~dead
                    want = np.zeros(n, po.XEVENT_DTYPE); want["poly_id"] = -1
                    if live.any(): want[live] = o.shoot(cur[live], excl1=excl[live])[0]
                    bad = same(ev, want); checks += 1
                    if bad: print("MISMATCH seed %d voxel D=%d n=%d kernel=%s bounce %d" % (seed, D, n, kern, b), bad); return 1
                    alive_ = (want["hit"] == 1) & live
                    cur = po.reflect_batch(To, cur, want)
                    excl = np.where(alive_, want["poly_id"], -2).astype(np.int32)
                    dead |= ~alive_
            g.set_option("voxel_kernel", 0)
            tmax = np.abs(rng.normal(0, 1, n)) * float(np.nanmedian(ref["t"][ref["hit"] == 1])) if (ref["hit"] == 1).any() else np.ones(n)
            occ = g.Occluded_batch(rays, t_max=tmax)[0]
            want_occ = (ref["hit"] == 1) & (ref["t"] < tmax); checks += 1
            if not np.array_equal(np.asarray(occ, bool), want_occ): print("MISMATCH seed %d occlusion D=%d n=%d" % (seed, D, n)); return 1
        if seed % 2 == 0:
            # KDTree: K3d (hare_kdtree_dense, round 5) and the one-ray-per-lane kernel, plain and with exclusions, boxes on and off
            kdepth = int(rng.integers(0, 14))
            kd, ko = H.KDTree([T], kdepth, maxp), po.KDTree([To], kdepth, maxp)
            m = min(n, 800)
            kref, krefx = ko.shoot(rays[:m])[0], ko.shoot(rays[:m], excl1=e1[:m], excl2=e2[:m])[0]
            for kern in (2, 1):
                kd.set_option("kdtree_kernel", kern)
                for tight in ((1, 0) if seed % 4 == 0 else (1,)):
                    kd.set_option("octree_tight", tight)
                    for what, got, want in (("plain", kd.Shoot_batch(rays[:m])[0], kref),
                                            ("excl", kd.Shoot_batch(rays[:m], poly_origin1=e1[:m], poly_origin2=e2[:m])[0], krefx)):
                        bad = same(got, want); checks += 1
                        if bad: print("MISMATCH seed %d kd %d/%d kernel %d tight %d %s" % (seed, kdepth, maxp, kern, tight, what), bad); return 1
            kd.set_option("kdtree_kernel", 0); kd.set_option("octree_tight", 1)
            if not occl_check(kd, "kd %d/%d" % (kdepth, maxp), kref, rays[:m], None, None): return 1
            if not occl_check(kd, "kd %d/%d excl" % (kdepth, maxp), krefx, rays[:m], e1[:m], e2[:m]): return 1
        if seed % 3 == 0:
            # the pool kernel through the cost order (voxel_order forced on: the rule only takes batches of 1.5M primary rays)
            g.set_option("voxel_order", 2)
            for what, got, want in (("plain", g.Shoot_batch(rays)[0], ref), ("excl", g.Shoot_batch(rays, poly_origin1=e1, poly_origin2=e2)[0], refx)):
                bad = same(got, want); checks += 1
                if bad: print("MISMATCH seed %d voxel D=%d n=%d cost order %s" % (seed, D, n, what), bad); return 1
            g.set_option("voxel_order", 1)
        print("seed %d clean (D=%d n=%d octree %d/%d), %d comparisons so far, %.0f s" % (seed, D, n, depth, maxp, checks, time.time() - t0), flush=True)
    print("CLEAN: seeds %d..%d, %d comparisons of full X_Event arrays, %.0f s" % (lo, hi - 1, checks, time.time() - t0))
    return 0


if __name__ == "__main__":
    sys.exit(main())
