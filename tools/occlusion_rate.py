"""Rate of the occlusion predicate with and without events: 1M burst rays, t_max = 0.25 / 0.5 / 1 / inf x the
scene's mean free path (mean closest-hit distance of the burst), device-resident (HIP events on the launch stream) and from host
buffers.  Flags are checked against the closest-hit records on every ray.  usage: python tools/occlusion_rate.py [hall|cathedral] [domain]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hare_amd as H
from hare_amd import capi

scene = sys.argv[1] if len(sys.argv) > 1 else "hall"
D = int(sys.argv[2]) if len(sys.argv) > 2 else (64 if scene == "hall" else 128)
n = int(os.environ.get("RAYS", 1 << 20))
mesh = H.scenes.SCENES[scene](); T = H.Topology(mesh.verts, mesh.nverts)
st = torch.cuda.current_stream().cuda_stream
rays = H.scenes.burst_rays(n, mesh.size)
d_rays = torch.from_numpy(rays).cuda()
d_ev = torch.empty(n * 56, dtype=torch.uint8, device="cuda")
d_occ = torch.zeros(n, dtype=torch.int32, device="cuda")


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for kind, g in (("voxel D=%d" % D, H.Voxel_Grid([T], D)), ("octree 8/16", H.Octree([T], 8, 16)), ("kdtree 16/8", H.KDTree([T], 16, 8))):
    ev, _ = g.Shoot_batch(rays)
    mfp = float(ev["t"][ev["hit"] != 0].mean())
    ms_shoot = timed(lambda: g.shoot_device(n, d_rays.data_ptr(), d_ev.data_ptr(), stream=st))
    print("%s %s: closest-hit Shoot %.3f ms (%.0f Mrays/s); mean free path %.2f m" % (scene, kind, ms_shoot, n / ms_shoot / 1e3, mfp), flush=True)
    for f in (0.25, 0.5, 1.0, float("inf")):
        tmax = np.full(n, mfp * f)
        d_tmax = torch.from_numpy(tmax).cuda()
        want = ((ev["hit"] != 0) & (ev["t"] < tmax)).astype(np.int32)
        ms_full = timed(lambda: g.occluded_device(n, d_rays.data_ptr(), d_ev.data_ptr(), d_occ.data_ptr(), d_tmax=d_tmax.data_ptr(), stream=st))
        ok_full = np.array_equal(d_occ.cpu().numpy(), want)
        d_occ.zero_()
        ms_flag = timed(lambda: g.occluded_device(n, d_rays.data_ptr(), 0, d_occ.data_ptr(), d_tmax=d_tmax.data_ptr(), stream=st))
        ok_flag = np.array_equal(d_occ.cpu().numpy(), want)
        if f == float("inf"):      # ... and the same question asked WITHOUT a t_max array (the octree then gets hare_octree_occl_any)
            d_occ.zero_()
            ms_none = timed(lambda: g.occluded_device(n, d_rays.data_ptr(), 0, d_occ.data_ptr(), stream=st))
            print("  t_max = NULL      : flags only %.3f ms (%5.0f Mrays/s), flags equal: %s" % (ms_none, n / ms_none / 1e3, np.array_equal(d_occ.cpu().numpy(), want)), flush=True)
        import ctypes as C
        occ = np.zeros(n, np.int32); evh = np.zeros(n, capi.XEVENT_DTYPE); ctr = capi.Counters()

        def host(events):      # the C-ABI call on preallocated host arrays, best of 3 (the first call sizes the staging buffers)
            best = 1e9
            for k in range(4):
                t0 = time.perf_counter()
                capi.check(capi.lib.hare_occluded_batch(g._h, g._kind, 0, n, rays.ctypes.data, None, None, tmax.ctypes.data, 0, occ.ctypes.data,
                                                        evh.ctypes.data if events else None, C.addressof(ctr)))
                if k: best = min(best, time.perf_counter() - t0)
            return best
        h_full = host(True); h_flag = host(False)
        print("  t_max = %4s x mfp: occluded %5.1f %% | with events %.3f ms (%5.0f Mrays/s) | flags only %.3f ms (%5.0f Mrays/s, x%.2f) | "
              "from host buffers %.0f -> %.0f Mrays/s | flags equal: %s %s %s"
              % (f, 100 * want.mean(), ms_full, n / ms_full / 1e3, ms_flag, n / ms_flag / 1e3, ms_full / ms_flag, n / h_full / 1e6, n / h_flag / 1e6,
                 ok_full, ok_flag, np.array_equal(occ, want)), flush=True)
