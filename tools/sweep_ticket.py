"""Developer sweep: ticket size (HARE_TICKET) x batch size for the persistent voxel kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hare_amd as H
D = int(os.environ.get("DOMAIN", 64))
mesh = H.scenes.hall(); g = H.Voxel_Grid([H.Topology(mesh.verts, mesh.nverts)], D)
st = torch.cuda.current_stream().cuda_stream
sizes = [int(x) for x in os.environ.get("SIZES", "65536,262144,1048576,2097152,4194304,8388608,16777216").split(",")]
tickets = [int(x) for x in sys.argv[1:]] or [16, 32, 64, 128, 256]
for N in sizes:
    rays = H.scenes.burst_rays(N, mesh.size)
    dr = torch.from_numpy(rays).cuda(); out = torch.empty(N * 56, dtype=torch.uint8, device="cuda")
    row = []
    for t in tickets:
        g.set_option("ticket_rays", t)
        K = max(5, min(40, (1 << 25) // N))
        for _ in range(2): g.shoot_device(N, dr.data_ptr(), out.data_ptr(), stream=st)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(K): g.shoot_device(N, dr.data_ptr(), out.data_ptr(), stream=st)
        e1.record(); torch.cuda.synchronize()
        row.append(N / (e0.elapsed_time(e1) / K) / 1e3)
    print("n=%9d  " % N + "  ".join("t%d: %.0f" % (t, r) for t, r in zip(tickets, row)), flush=True)
