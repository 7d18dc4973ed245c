"""Developer sweep of the persistent kernel's scheduling knobs (HARE_TUNE) on the bench workload."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hare_amd as H
D = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = 1 << 20
mesh = H.scenes.hall(); g = H.Voxel_Grid([H.Topology(mesh.verts, mesh.nverts)], D)
rays = H.scenes.burst_rays(N, mesh.size)
dr = torch.from_numpy(rays).cuda(); out = torch.empty(N * 56, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def run(K=20):
    for _ in range(3): g.shoot_device(N, dr.data_ptr(), out.data_ptr(), stream=st)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K): g.shoot_device(N, dr.data_ptr(), out.data_ptr(), stream=st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K
configs = [tuple(int(x) for x in a.split(",")) for a in sys.argv[2:]] or \
    [(s, r, c, 4, e) for s in (2, 4, 8) for r in (8, 16) for c in (64, 128) for e in (8, 16, 24, 32, 48)]
res = []
for cfg in configs:
    os.environ["HARE_TUNE"] = ",".join(str(x) for x in cfg)
    ms = run()
    res.append((ms, cfg)); print(cfg, "%.3f ms  %.0f Mrays/s" % (ms, N / ms / 1e3), flush=True)
res.sort(); print("best:", res[:5])
