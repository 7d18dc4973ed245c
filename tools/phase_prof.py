"""Developer tool: per-phase cycle shares and lane occupancy of the persistent voxel kernel (profiling build: the
production kernel + cycle stamps kept in LDS, same register footprint and occupancy).

    python tools/phase_prof.py [D]                      # the burst
    SCENE=cathedral BOUNCE=4 python tools/phase_prof.py 80   # the ray front after 4 specular bounces (poly_origin set)
The profiling build exists for triangle scenes on grids with one occupancy bit per voxel (D <= 80)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["HARE_DEV"] = "1"                 # the profiling flag is a developer bit
os.environ["HARE_VOXEL_KERNEL"] = "persist"
import numpy as np, torch
import hare_amd as H
D = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = 1 << 20
KB = int(os.environ.get("BOUNCE", 0))
mesh = H.scenes.SCENES[os.environ.get("SCENE", "hall")](); g = H.Voxel_Grid([H.Topology(mesh.verts, mesh.nverts)], D)
rays = H.scenes.burst_rays(N, mesh.size)
dr = torch.from_numpy(rays).cuda(); out = torch.empty(N * 56, dtype=torch.uint8, device="cuda")
de = torch.full((N,), -1, dtype=torch.int32, device="cuda")
buf = torch.zeros(8 + 17, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for b in range(KB):
    g.shoot_device(N, dr.data_ptr(), out.data_ptr(), d_excl1=de.data_ptr(), stream=st, flags=H.capi.SHOOT_RETIRED_RAYS)
    g.reflect_device(N, dr.data_ptr(), out.data_ptr(), de.data_ptr(), stream=st)
torch.cuda.synchronize()
print("%s, D = %d, %s" % (os.environ.get("SCENE", "hall"), D, "burst" if KB == 0 else "ray front after %d bounces" % KB))
cw = torch.zeros(8, dtype=torch.int64, device="cuda")
g.shoot_device(N, dr.data_ptr(), out.data_ptr(), d_excl1=de.data_ptr(), d_counters=cw.data_ptr(), stream=st,
               flags=H.capi.SHOOT_RETIRED_RAYS | H.capi.SHOOT_COUNT_WORK)
torch.cuda.synchronize()
c = cw.cpu().numpy()
print("  reference algorithm per ray: cells %.1f entries %.1f intersect calls %.1f (hits %d of %d)" % (c[2] / N, c[3] / N, c[4] / N, c[1], c[0]))
for tune in ["production constants"]:      # the profiling build uses the production kernel's compile-time knobs
    buf.zero_()
    g.shoot_device(N, dr.data_ptr(), out.data_ptr(), d_excl1=de.data_ptr(), d_counters=buf.data_ptr(), stream=st,
                   flags=0x4000 | H.capi.SHOOT_RETIRED_RAYS)
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
