"""Developer measurement: hare_bounce_device skips retired rays instead of packing the survivors -- what does a cast
cost once (nearly) every ray is dead?  Scene: one tessellated wall (the hall's floor, 18 942 triangles, D = 64); a burst from above: half the
rays hit it in cast 0, their reflections fly away, so casts 2 ... B-1 find only retired rays.  Reported: the loop at B = 2 and B = 8 on the
same rays; (t8 - t2) / 6 is the cost of a cast over n retired rays -- against a cast of live rays.
    RAYS=4194304 python tools/bounce_open_scene.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hare_amd as H
from hare_amd.scenes import _patch, _finish, _n

N = int(os.environ.get("RAYS", 1 << 22))
edge = 83.0 / 256.0
mesh = _finish("one-wall", [_patch([0, 0, 0], [40.0, 0, 0], [0, 25.0, 0], _n(40.0, edge), _n(25.0, edge))], (40.0, 25.0, 18.0))
g = H.Voxel_Grid([H.Topology(mesh.verts, mesh.nverts)], 64)
rays0 = torch.from_numpy(H.scenes.burst_rays(N, mesh.size)).cuda()
rays = rays0.clone(); work = torch.zeros(2 * N, dtype=torch.int32, device="cuda"); out = torch.empty(N * 56, dtype=torch.uint8, device="cuda")
ctr = torch.zeros(8 * 8, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream


def loop(B):
    rays.copy_(rays0)
    g.bounce_device(N, rays.data_ptr(), B, work.data_ptr(), d_events_last=out.data_ptr(), d_counters_per_cast=ctr.data_ptr(), stream=st)


res = {}
for B in (1, 2, 8):
    ctr.zero_(); loop(B); torch.cuda.synchronize()
    per = ctr.cpu().numpy().reshape(8, 8)[:B, :2].tolist()
    best = 1e9
    for rep in range(4):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); loop(B); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    res[B] = best
    print("B = %d: %.3f ms per loop; {rays started, hits} per cast of the first run: %s" % (B, best, per), flush=True)
print("n = %d rays, kernel %s: a cast of live rays %.3f ms; a cast over n RETIRED rays (t8 - t2) / 6 = %.3f ms = %.1f us per million rays"
      % (N, g.kernel_name(N), res[1], (res[8] - res[2]) / 6, (res[8] - res[2]) / 6 * 1e3 / (N / 1e6)))
