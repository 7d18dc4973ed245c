"""Developer measurement: consecutive batches on ONE stream against the same batches alternating over TWO (or more)
streams.  A persistent launch ends with a drain (waves finish at different times) and starts with a ramp; on two streams the next
batch's workgroups take the CUs the previous batch's waves leave, so one launch's drain overlaps the other's ramp.  What a streaming
caller of hare_shoot_device can do today -- results do not depend on it.  usage: python tools/two_stream.py [streams ...]"""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hare_amd as H

NS = [int(a) for a in sys.argv[1:]] or [1, 2, 3]
hall = H.scenes.hall(); Th = H.Topology(hall.verts, hall.nverts)
cases = [("C2 hall voxel D=64, 1M rays (K1p)", hall, H.Voxel_Grid([Th], 64), 1 << 20, 0),
         ("hall voxel D=64, 1M rays, pool kernel forced (K1q)", hall, H.Voxel_Grid([Th], 64), 1 << 20, 2),
         ("C3 hall octree 8/16, 1M rays (K2p)", hall, H.Octree([Th], 8, 16), 1 << 20, 0)]
if os.environ.get("CATHEDRAL", "1") == "1":
    cath = H.scenes.cathedral(); Tc = H.Topology(cath.verts, cath.nverts)
    cases.append(("C4 shard cathedral voxel D=128, 2M rays (K1q)", cath, H.Voxel_Grid([Tc], 128), 1 << 21, 0))
for name, mesh, g, n, vk in cases:
    if vk: g.set_option("voxel_kernel", vk)
    rays = H.scenes.burst_rays(n, mesh.size)
    K = 12
    sets = [(torch.from_numpy(rays).cuda(), torch.empty(n * 56, dtype=torch.uint8, device="cuda")) for _ in range(4)]
    ref = None
    print(name, g.kernel_name(n), flush=True)
    for ns in NS:
        streams = [torch.cuda.Stream() for _ in range(ns)]
        def run(reps):
            for k in range(reps):
                st = streams[k % ns]; r, o = sets[k % len(sets)]
                g.shoot_device(n, r.data_ptr(), o.data_ptr(), stream=st.cuda_stream)
        run(4); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream())
        for s in streams: s.wait_event(e0)
        run(K)
        evs = []
        for s in streams:
            e = torch.cuda.Event(); e.record(s); evs.append(e)
        for e in evs: torch.cuda.current_stream().wait_event(e)
        e1.record(torch.cuda.current_stream()); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / K
        crc = [zlib.crc32(o.cpu().numpy().tobytes()) for _, o in sets]
        if ref is None: ref = crc
        print("   %d stream(s): %.3f ms per batch, %.0f Mrays/s; events identical to the one-stream run: %s" % (ns, ms, n / ms / 1e3, crc == ref), flush=True)
