import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hare_amd as H
N = 1 << 20
mesh = H.scenes.hall(); g = H.Octree([H.Topology(mesh.verts, mesh.nverts)], 8, 16)
rays = H.scenes.burst_rays(N, mesh.size)
dr = torch.from_numpy(rays).cuda(); out = torch.empty(N * 56, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def run(K=3):
    g.shoot_device(N, dr.data_ptr(), out.data_ptr(), stream=st); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K): g.shoot_device(N, dr.data_ptr(), out.data_ptr(), stream=st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K
for cfg in sys.argv[1:]:
    os.environ["HARE_TUNE"] = cfg
    ms = run(); print(cfg, "%.2f ms %.0f Mrays/s" % (ms, N / ms / 1e3), flush=True)
