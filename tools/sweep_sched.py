"""Developer sweep of the staggered-retirement draw budgets (HARE_BUDGET=f1,f2,f3: fractions of the fair share of ticket chunks) on the bench workload.
The first configuration should be "tickets" (reference output); every other output is compared with it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hare_amd as H
N = int(os.environ.get("RAYS", 1 << 20))
D = int(os.environ.get("DOMAIN", 64))
mesh = H.scenes.hall(); g = H.Voxel_Grid([H.Topology(mesh.verts, mesh.nverts)], D)
rays = H.scenes.burst_rays(N, mesh.size)
dr = torch.from_numpy(rays).cuda(); out = torch.empty(N * 56, dtype=torch.uint8, device="cuda")
ref = None
st = torch.cuda.current_stream().cuda_stream
def run(K=30):
    for _ in range(3): g.shoot_device(N, dr.data_ptr(), out.data_ptr(), stream=st)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K): g.shoot_device(N, dr.data_ptr(), out.data_ptr(), stream=st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K
res = []
for cfg in sys.argv[1:] or ["tickets"]:
    if cfg == "tickets": os.environ.pop("HARE_BUDGET", None)
    else: os.environ["HARE_BUDGET"] = cfg
    out.zero_()
    ms = run()
    o = out.cpu().numpy().copy()
    if ref is None: ref = o
    same = bool((o == ref).all())
    res.append((ms, cfg)); print(cfg, "%.4f ms  %.0f Mrays/s  same=%s" % (ms, N / ms / 1e3, same), flush=True)
res.sort(); print("best:", res[:5])
