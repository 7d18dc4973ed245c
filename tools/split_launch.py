"""Developer experiment (round 5): ONE batch cut into parts that run side by side on internal streams -- could hare_shoot_device hide part of
a launch's ramp and drain inside a single call?  (tools/two_stream.py is the other question: CONSECUTIVE batches on two streams.)  Every
batch is fenced: its parts start after the previous batch has finished completely and the batch is finished when all of its parts are, which
is what a stream-ordered call that forked and joined inside would look like to its caller.  Events compared (CRC) with the one-launch run's.
    python tools/split_launch.py
"""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hare_amd as H

hall = H.scenes.hall(); Th = H.Topology(hall.verts, hall.nverts)
cases = [("C2 hall voxel D=64, 1M rays", hall, H.Voxel_Grid([Th], 64), 1 << 20),
         ("C3 hall octree 8/16, 1M rays", hall, H.Octree([Th], 8, 16), 1 << 20),
         ("C3 hall octree 8/16, 262 144 rays", hall, H.Octree([Th], 8, 16), 1 << 18),
         ("hall kd 16/8, 1M rays", hall, H.KDTree([Th], 16, 8), 1 << 20)]
SPLITS = [(1.0,), (0.5, 0.5), (0.75, 0.25), (0.85, 0.15), (0.9, 0.1), (0.6, 0.3, 0.1), (0.25, 0.25, 0.25, 0.25)]
main = torch.cuda.current_stream()
pool = [torch.cuda.Stream() for _ in range(4)]
for name, mesh, g, n in cases:
    rays = H.scenes.burst_rays(n, mesh.size)
    d_r = torch.from_numpy(rays).cuda(); d_o = torch.empty(n * 56, dtype=torch.uint8, device="cuda")
    print(name, g.kernel_name(n), flush=True)
    ref = None
    for split in SPLITS:
        cuts = [0]
        for f in split[:-1]: cuts.append(min(n, (cuts[-1] + int(n * f) + 63) // 64 * 64))
        cuts.append(n)

        def batch():
            if len(split) == 1:
                g.shoot_device(n, d_r.data_ptr(), d_o.data_ptr(), stream=main.cuda_stream); return
            e = torch.cuda.Event(); e.record(main)
            for k in range(len(split)):
                a, b = cuts[k], cuts[k + 1]
                if b <= a: continue
                s = pool[k]; s.wait_event(e)
                g.shoot_device(b - a, d_r.data_ptr() + a * 48, d_o.data_ptr() + a * 56, stream=s.cuda_stream)
                d = torch.cuda.Event(); d.record(s); main.wait_event(d)
        K = 12
        for _ in range(3): batch()
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(main)
            for _ in range(K): batch()
            e1.record(main); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / K)
        crc = zlib.crc32(d_o.cpu().numpy().tobytes())
        if ref is None: ref, base = crc, best
        print("   parts %-24s %.4f ms (%+.1f %%) %.0f Mrays/s%s" % ("/".join("%g" % f for f in split), best, 100 * (best / base - 1), n / best / 1e3,
                                                                   "" if crc == ref else "  EVENTS DIFFER"), flush=True)
