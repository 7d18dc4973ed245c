"""Developer sweep: rays per ticket and static first chunk of the octree kernel K2p (scene options "ticket_rays", "k2p_static_rays"; 0 = the host rules)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hare_amd as H
mesh = H.scenes.hall(); g = H.Octree([H.Topology(mesh.verts, mesh.nverts)], 8, 16)
st = torch.cuda.current_stream().cuda_stream
for N in [int(x) for x in os.environ.get("RAYS", "1048576,4194304").split(",")]:
    rays = H.scenes.burst_rays(N, mesh.size)
    dr = torch.from_numpy(rays).cuda(); out = torch.empty(N * 56, dtype=torch.uint8, device="cuda")
    for t in [int(x) for x in os.environ.get("TICKETS", "0,16,32,64,128").split(",")]:
        for sr in [int(x) for x in os.environ.get("STATICS", "0,64,128").split(",")]:
            g.set_option("ticket_rays", t); g.set_option("k2p_static_rays", sr)
            best = 1e9
            for rep in range(2):
                for _ in range(2): g.shoot_device(N, dr.data_ptr(), out.data_ptr(), stream=st)
                torch.cuda.synchronize()
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(); K = 6 if N < 2e6 else 3
                for _ in range(K): g.shoot_device(N, dr.data_ptr(), out.data_ptr(), stream=st)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / K)
            print("n=%d ticket=%d static=%d: %.4f ms %.0f Mrays/s" % (N, t, sr, best, N / best / 1e3), flush=True)
