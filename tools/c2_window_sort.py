"""Developer experiment (round 5): rays of similar COST together, the burst's locality kept -- sorted by the walk-length estimate inside
windows of W consecutive rays (tools/c2_heavy_first.py found -7 % at W = 8192 where any global heavy-first order gave -1.5 %).

  * window sizes, descending / ascending, the key quantised to a few classes (what a counting sort in LDS would produce);
  * PHYSICAL permutation (rays uploaded in the new order; the events un-permuted on the host) against the INDIRECT one the library
    could do itself: rays, events and exclusions stay where the caller has them and K1q takes the rays in the order of a device array
    (ShootIO::order through the developer option dev_order_ptr);
  * BOUNCE=k: the front of bounce k in the cathedral instead of the burst.

Events are compared (CRC) with the unpermuted run's in every case.
    SCENE=hall DOMAIN=64 RAYS=1048576 [BOUNCE=0] python tools/c2_window_sort.py
"""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["HARE_DEV"] = "1"
import numpy as np, torch
import hare_amd as H


def main():
    D = int(os.environ.get("DOMAIN", 64)); KB = int(os.environ.get("BOUNCE", 0))
    mesh = H.scenes.SCENES[os.environ.get("SCENE", "hall")]()
    KIND = os.environ.get("KIND", "voxel")          # octree / kdtree: the same estimate (a virtual grid of D voxels a side), physical permutations only
    T0 = H.Topology(mesh.verts, mesh.nverts)
    g = H.Voxel_Grid([T0], D) if KIND == "voxel" else (H.Octree([T0], 8, 16) if KIND == "octree" else H.KDTree([T0], 16, 8))
    st = torch.cuda.current_stream().cuda_stream
    V = np.asarray(mesh.verts).reshape(-1, 4, 3)[:, :3, :].reshape(-1, 3)
    lo, hi = V.min(0), V.max(0)
    vd = (hi - lo) / D
    windows = [int(x) for x in os.environ.get("WINDOWS", "1024,2048,4096,8192,16384,32768").split(",")]
    for N in [int(x) for x in os.environ.get("RAYS", "1048576").split(",")]:
        rays = H.scenes.burst_rays(N, mesh.size)
        excl = None
        if KB > 0:
            d_rays = torch.from_numpy(rays).cuda(); d_out = torch.empty(N * 56, dtype=torch.uint8, device="cuda")
            d_excl = torch.full((N,), -1, dtype=torch.int32, device="cuda")
            for b in range(KB):
                g.shoot_device(N, d_rays.data_ptr(), d_out.data_ptr(), d_excl1=d_excl.data_ptr(), stream=st, flags=H.capi.SHOOT_RETIRED_RAYS)
                g.reflect_device(N, d_rays.data_ptr(), d_out.data_ptr(), d_excl.data_ptr(), stream=st)
            torch.cuda.synchronize()
            rays = d_rays.cpu().numpy().reshape(N, 6).copy(); excl = d_excl.cpu().numpy().copy()
        o, d = rays[:, :3], rays[:, 3:]
        with np.errstate(divide="ignore", invalid="ignore"):
            t0 = (lo - o) / d; t1 = (hi - o) / d
            t_exit = np.nanmin(np.maximum(t0, t1), 1)
            cells = np.nan_to_num((np.abs(d) * t_exit[:, None] / vd).sum(1), nan=0.0, posinf=0.0)
        fl = H.capi.SHOOT_RETIRED_RAYS if excl is not None else 0

        def timed(dr, out, de, K=10):
            ep = de.data_ptr() if de is not None else 0
            for _ in range(3): g.shoot_device(N, dr.data_ptr(), out.data_ptr(), d_excl1=ep, stream=st, flags=fl)
            torch.cuda.synchronize()
            best = 1e9
            for rep in range(3):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(K): g.shoot_device(N, dr.data_ptr(), out.data_ptr(), d_excl1=ep, stream=st, flags=fl)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / K)
            return best

        def run_physical(perm):
            dr = torch.from_numpy(np.ascontiguousarray(rays[perm])).cuda(); out = torch.empty(N * 56, dtype=torch.uint8, device="cuda")
            de = None if excl is None else torch.from_numpy(np.ascontiguousarray(excl[perm])).cuda()
            if KIND == "voxel": g.set_option("dev_order_ptr", 0)
            t = timed(dr, out, de)
            ev = out.cpu().numpy().reshape(N, 56); back = np.empty_like(ev); back[perm] = ev
            return t, zlib.crc32(back.tobytes())

        dr0 = torch.from_numpy(rays).cuda(); out0 = torch.empty(N * 56, dtype=torch.uint8, device="cuda")
        de0 = None if excl is None else torch.from_numpy(excl).cuda()

        def run_indirect(perm):
            order = torch.from_numpy(perm.astype(np.uint32)).cuda()
            g.set_option("dev_order_ptr", order.data_ptr())
            out0.zero_()
            t = timed(dr0, out0, de0)
            g.set_option("dev_order_ptr", 0)
            return t, zlib.crc32(out0.cpu().numpy().tobytes())

        t_base, c0 = run_physical(np.arange(N))
        print("%s D=%d n=%d bounce %d kernel %s: as given %.4f ms = %.0f M/s; cells-to-exit mean %.1f p99 %.1f"
              % (mesh.name, D, N, KB, g.kernel_name(N), t_base, N / t_base / 1e3, cells.mean(), np.percentile(cells, 99)), flush=True)

        def window_perm(W, key):
            perm = np.arange(N)
            for a in range(0, N, W):
                seg = perm[a:a + W]
                perm[a:a + W] = seg[np.argsort(key[seg], kind="stable")]
            return perm

        def report(label, perm):
            tp, cp = run_physical(perm)
            if KIND != "voxel":
                print("   %-44s physical %.4f ms (%+.1f %%)%s" % (label, tp, 100 * (tp / t_base - 1), "" if cp == c0 else " EVENTS DIFFER"), flush=True)
                return
            ti, ci = run_indirect(perm)
            print("   %-44s physical %.4f ms (%+.1f %%)%s | through order[] %.4f ms (%+.1f %%)%s"
                  % (label, tp, 100 * (tp / t_base - 1), "" if cp == c0 else " EVENTS DIFFER", ti, 100 * (ti / t_base - 1), "" if ci == c0 else " EVENTS DIFFER"), flush=True)

        report("identity (the cost of the indirection itself)", np.arange(N))
        for W in windows:
            report("window %6d, descending" % W, window_perm(W, -cells))
        report("window %6d, ascending" % 8192, window_perm(8192, cells))
        # quantised keys: 4 / 8 / 16 classes of equal WIDTH between the window's own min and max would need two passes; classes of a fixed
        # width in voxels (what one pass can do) are tried instead
        for W, step in ((8192, 4.0), (8192, 8.0), (8192, 16.0), (4096, 8.0), (16384, 8.0)):
            report("window %6d, classes of %2.0f voxels, desc." % (W, step), window_perm(W, -np.floor(cells / step)))
        t, c = run_physical(np.random.default_rng(0).permutation(N))
        print("   %-44s physical %.4f ms (%+.1f %%)%s" % ("random permutation", t, 100 * (t / t_base - 1), "" if c == c0 else " EVENTS DIFFER"), flush=True)


if __name__ == "__main__":
    main()
