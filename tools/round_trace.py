"""Developer tool: per-ROUND trace of the pool kernel K1q (flag 0x1000): one wave in 256 stamps every round with the 100 MHz clock
and its queue lengths {walk, cull, exact, pend}.  Answers: how long is a round in the steady state and in the drain, how full are
the phases, how many rounds does the drain take.  usage: [SCENE=hall|cathedral] [DOMAIN=64] [RAYS=1048576] [BOUNCE=k] python tools/round_trace.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
os.environ["HARE_DEV"] = "1"
import hare_amd as H
N = int(os.environ.get("RAYS", 1 << 20)); D = int(os.environ.get("DOMAIN", 64)); BOUNCE = int(os.environ.get("BOUNCE", 0))
mesh = getattr(H.scenes, os.environ.get("SCENE", "hall"))()
g = H.Voxel_Grid([H.Topology(mesh.verts, mesh.nverts)], D)
g.set_option("voxel_kernel", 2)
rays = H.scenes.burst_rays(N, mesh.size)
dr = torch.from_numpy(rays).cuda(); out = torch.empty(N * 56, dtype=torch.uint8, device="cuda")
excl = torch.full((N,), -1, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for b in range(BOUNCE):      # the ray front after BOUNCE specular bounces
    g.shoot_device(N, dr.data_ptr(), out.data_ptr(), d_excl1=excl.data_ptr(), stream=st, flags=H.capi.SHOOT_RETIRED_RAYS)
    g.reflect_device(N, dr.data_ptr(), out.data_ptr(), excl.data_ptr(), stream=st)
TW = 16
buf = torch.zeros(8 + 32 + 4 * 4096 + TW * 1024 + 4096 * 48, dtype=torch.int64, device="cuda")
for rep in range(2):
    buf.zero_()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    g.shoot_device(N, dr.data_ptr(), out.data_ptr(), d_excl1=excl.data_ptr(), d_counters=buf.data_ptr(), stream=st,
                   flags=0x1000 | H.capi.SHOOT_RETIRED_RAYS)
    e1.record(); torch.cuda.synchronize()
print("kernel %s, %d rays, %.3f ms with the trace on" % (g.kernel_name(N), N, e0.elapsed_time(e1)))
allb = buf.cpu().numpy()
tr = allb[8 + 32 + 4 * 4096:8 + 32 + 4 * 4096 + TW * 1024].astype(np.uint64).reshape(TW, 1024)
dr_all = allb[8 + 32 + 4 * 4096 + TW * 1024:].astype(np.uint64).reshape(4096, 48)
# the waves whose drain ends last: their rounds after the tickets ran dry
tt = (dr_all >> np.uint64(34)).astype(np.int64)
lastt = tt.max(1); firstt = np.where(tt > 0, tt, np.int64(1) << 62).min(1)
livew = np.nonzero(lastt > 0)[0]
tmin_all = firstt[livew].min()
order = livew[np.argsort(-lastt[livew])][:10]
print("drain length per wave (first dry round .. last traced round): mean %.0f p50 %.0f p90 %.0f max %.0f us; drain rounds mean %.1f max %d"
      % (((lastt - firstt)[livew] / 100.0).mean(), np.median((lastt - firstt)[livew]) / 100.0, np.percentile((lastt - firstt)[livew], 90) / 100.0,
         (lastt - firstt)[livew].max() / 100.0, (tt[livew] > 0).sum(1).mean(), (tt[livew] > 0).sum(1).max()))
for w in order:
    v = dr_all[w]; v = v[v != 0]
    t = ((v >> np.uint64(34)).astype(np.int64) - tmin_all) / 100.0
    q = np.stack([((v >> np.uint64(s)) & np.uint64(255)).astype(int) for s in (0, 8, 16, 24)], 1)
    print("late wave %4d (block %d): drain rounds (us since the first wave ran dry: walk/cull/exact/pend):" % (w, w // 12),
          " ".join("%.0f:%d/%d/%d/%d" % (t[k], *q[k]) for k in range(len(v))))
t_all0 = None
for w in range(TW):
    v = tr[w]; v = v[v != 0]
    if len(v) < 3: continue
    t = (v >> np.uint64(34)).astype(np.int64); t = (t - t[0]) / 100.0          # microseconds since this wave's first round
    dr_ = ((v >> np.uint64(32)) & np.uint64(1)).astype(int)
    q = np.stack([((v >> np.uint64(s)) & np.uint64(255)).astype(int) for s in (0, 8, 16, 24)], 1)
    dt = np.diff(t)
    steady = dr_[:-1] == 0
    i_dry = int(np.argmax(dr_)) if dr_.any() else len(v)
    print("wave %2d: %3d rounds, life %.0f us; dry at round %d (%.0f us); round time steady mean %.2f us (p50 %.2f p90 %.2f), drain mean %.2f us (p50 %.2f, p90 %.2f); "
          "rays in pool steady mean %.0f" % (w, len(v), t[-1], i_dry, t[min(i_dry, len(t) - 1)], dt[steady].mean() if steady.any() else 0,
          np.median(dt[steady]) if steady.any() else 0, np.percentile(dt[steady], 90) if steady.any() else 0,
          dt[~steady].mean() if (~steady).any() else 0, np.median(dt[~steady]) if (~steady).any() else 0, np.percentile(dt[~steady], 90) if (~steady).any() else 0,
          q[:i_dry].sum(1).mean() if i_dry else 0))
    if w < 3:
        print("   drain rounds (us since dry: walk/cull/exact/pend):", " ".join("%.0f:%d/%d/%d/%d" % (t[k] - t[min(i_dry, len(t) - 1)], *q[k]) for k in range(i_dry, len(v))))
        print("   first rounds:", " ".join("%.0f:%d/%d/%d/%d" % (t[k], *q[k]) for k in range(min(14, len(v)))))
