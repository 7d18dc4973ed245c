"""Developer sweep: rays per ticket of the pool kernel (scene option "ticket_rays"; 0 = the host rule) at 1M and 4M rays, hall D = 64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hare_amd as H
mesh = getattr(H.scenes, os.environ.get("SCENE", "hall"))(); g = H.Voxel_Grid([H.Topology(mesh.verts, mesh.nverts)], int(os.environ.get("DOMAIN", 64)))
st = torch.cuda.current_stream().cuda_stream
for N in [int(x) for x in os.environ.get('RAYS', '1048576,4194304').split(',')]:
    rays = H.scenes.burst_rays(N, mesh.size)
    dr = torch.from_numpy(rays).cuda(); out = torch.empty(N * 56, dtype=torch.uint8, device="cuda")
    for t in [int(x) for x in os.environ.get('TICKETS', '0,16,24,32,48,64,96,128').split(',')]:
        g.set_option("ticket_rays", t)
        g.set_option("k1p_static_rays", int(os.environ.get("STATIC", 0)))
        best = 1e9
        for rep in range(3):
            for _ in range(3): g.shoot_device(N, dr.data_ptr(), out.data_ptr(), stream=st)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); K = 20 if N < 2e6 else 8
            for _ in range(K): g.shoot_device(N, dr.data_ptr(), out.data_ptr(), stream=st)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / K)
        print("n=%d ticket_rays=%d: %.4f ms %.0f Mrays/s" % (N, t, best, N / best / 1e3), flush=True)
