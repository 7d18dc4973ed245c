"""Developer tool: per-phase cycle shares and lane occupancy of the persistent voxel kernel (profiling build: the
production kernel + cycle stamps kept in LDS, same register footprint and occupancy)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hare_amd as H
D = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = 1 << 20
mesh = H.scenes.hall(); g = H.Voxel_Grid([H.Topology(mesh.verts, mesh.nverts)], D)
rays = H.scenes.burst_rays(N, mesh.size)
dr = torch.from_numpy(rays).cuda(); out = torch.empty(N * 56, dtype=torch.uint8, device="cuda")
buf = torch.zeros(8 + 17, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for tune in ["production constants"]:      # the profiling build uses the production kernel's compile-time knobs
    buf.zero_()
    g.shoot_device(N, dr.data_ptr(), out.data_ptr(), d_counters=buf.data_ptr(), stream=st, flags=0x4000)
    torch.cuda.synchronize()
    p = buf.cpu().numpy()[8:].astype(np.float64)
    waves = p[16]; tot = p[0:5].sum()
    print("tune", tune, "waves", int(waves), "cycles/wave %.0f" % (tot / waves))
    names = ["prologue", "refill", "A walk", "B1 cull", "B2 exact"]
    for k in range(5): print("  %-9s %5.1f %%" % (names[k], 100 * p[k] / tot))
    print("  rounds/wave %.0f; alive lanes/round %.1f" % (p[5] / waves, p[14] / max(p[5], 1)))
    print("  A iters/round %.2f, lanes/iter %.1f | B1 runs/round %.2f lanes %.1f | B2 runs/round %.3f lanes %.1f | refills/wave %.1f lanes %.1f"
          % (p[6] / p[5], p[7] / max(p[6], 1), p[8] / p[5], p[9] / max(p[8], 1), p[10] / p[5], p[11] / max(p[10], 1), p[12] / waves, p[13] / max(p[12], 1)))
    print("  per ray: steps %.1f culls %.1f exact %.2f" % (p[7] / N, p[9] / N, p[11] / N))
