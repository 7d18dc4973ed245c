"""Developer: what K1q's rounds are made of (build: tools/build_variants.sh "qstats:-DHARE_K1Q_STATS=1"; run with
HARE_LIB=hare_amd/libhare_hip_qstats.so HARE_DEV=1 [BOUNCE=k] python tools/k1q_stats.py [rays] [scene] [domain]).  GPU box.
BOUNCE=k: the ray front after k specular bounces (a late cast of the bounce loop) instead of the primary burst."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import hare_amd as H

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
scene = sys.argv[2] if len(sys.argv) > 2 else "hall"
D = int(sys.argv[3]) if len(sys.argv) > 3 else 64
m = H.scenes.SCENES[scene]()
g = H.Voxel_Grid([H.Topology(m.verts, m.nverts)], D)
g.set_option("dev", 1)
rays = torch.from_numpy(H.scenes.burst_rays(n, m.size)).cuda()
out = torch.empty(n * 56, dtype=torch.uint8, device="cuda")
ctr = torch.zeros(8 + 64, dtype=torch.int64, device="cuda")
sp = torch.cuda.current_stream().cuda_stream
BOUNCE = int(os.environ.get("BOUNCE", 0))
excl = torch.full((n,), -1, dtype=torch.int32, device="cuda")
for b in range(BOUNCE):
    g.shoot_device(n, rays.data_ptr(), out.data_ptr(), d_excl1=excl.data_ptr(), stream=sp, flags=H.capi.SHOOT_RETIRED_RAYS)
    g.reflect_device(n, rays.data_ptr(), out.data_ptr(), excl.data_ptr(), stream=sp)
fl = H.capi.SHOOT_RETIRED_RAYS if BOUNCE else 0
g.shoot_device(n, rays.data_ptr(), out.data_ptr(), d_excl1=excl.data_ptr() if BOUNCE else 0, stream=sp, flags=fl)
ctr.zero_()
g.shoot_device(n, rays.data_ptr(), out.data_ptr(), d_excl1=excl.data_ptr() if BOUNCE else 0, d_counters=ctr.data_ptr(), stream=sp, flags=0x1000 | fl)
torch.cuda.synchronize()
c = ctr.cpu().numpy().astype(np.float64)
r = c[0]
names = ["set-up", "walk task", "cull task", "exact", "pend walk", "wide walk", "wide cull", "walk step (inside walk tasks)"]
print("rays", int(r), "rounds per ray x waves: %.3f wave-rounds per ray" % (c[8 + 16] / r))
for k, nm in enumerate(names):
    ex, ln = c[8 + 2 * k], c[8 + 2 * k + 1]
    if ex:
        print("%-32s executions per ray %.4f   lanes per execution %.1f   lane-tasks per ray %.2f" % (nm, ex / r, ln / ex, ln / r))
ex, ln = c[8 + 18], c[8 + 19]
if ex:
    print("%-32s executions per ray %.4f   lanes per execution %.1f   lane-tasks per ray %.2f" % ("pend step (inside pend walks)", ex / r, ln / ex, ln / r))
t = c[8 + 20: 8 + 30]
if t.sum() > 0:
    tn = names[:7] + ["walk step loop", "ticket draw (atomic on one address)", "round bookkeeping (queue choice, refill rule, pushes)"]
    print("share of a wave's time by phase (shader clock between phase starts):")
    for k in range(10):
        if t[k]:
            print("  %-56s %5.1f %%" % (tn[k], 100.0 * t[k] / t.sum()))
