"""Developer tool: per-wave timeline of the production octree kernel K2p (flag 0x2000): start, tickets dry, end -- how much of a
launch is its tail (a few rays scan thousands of leaf entries: tools prints their share)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["HARE_DEV"] = "1"
import numpy as np, torch
import hare_amd as H
mesh = H.scenes.hall(); g = H.Octree([H.Topology(mesh.verts, mesh.nverts)], 8, 16)
g.set_option("octree_kernel", int(os.environ.get("OCTREE_KERNEL", "4")))     # K2d at every size (the rule hands small batches to K2g, which has no stamps)
W = 4096
st = torch.cuda.current_stream().cuda_stream
for N in [int(x) for x in os.environ.get("RAYS", "1048576").split(",")]:
    rays = H.scenes.burst_rays(N, mesh.size)
    dr = torch.from_numpy(rays).cuda(); out = torch.empty(N * 56, dtype=torch.uint8, device="cuda")
    buf = torch.zeros(8 + 32 + 8 * W, dtype=torch.int64, device="cuda")
    for rep in range(2):
        buf.zero_()
        g.shoot_device(N, dr.data_ptr(), out.data_ptr(), d_counters=buf.data_ptr(), stream=st, flags=0x2000)
        torch.cuda.synchronize()
    tl = buf.cpu().numpy()[8 + 32:8 + 32 + 4 * W].reshape(W, 4).astype(np.float64)
    live = tl[:, 0] > 0
    t0 = tl[live, 0].min()
    dry = (tl[:, 1] - t0) / 100.0; end = (tl[:, 2] - t0) / 100.0
    span = end[live].max()
    has_dry = live & (tl[:, 1] > 0)
    e = np.sort(end[live])
    print("n=%d: %d waves, span %.0f us | tickets dry p10/50/90 %.0f/%.0f/%.0f | wave end p10/50/90/99/max %.0f/%.0f/%.0f/%.0f/%.0f"
          % (N, live.sum(), span, *np.percentile(dry[has_dry], [10, 50, 90]), *np.percentile(end[live], [10, 50, 90, 99]), span))
    print("   waves still running at 50/60/70/80/90/95 %% of the span: %s of %d; wave-time after the median wave's end: %.1f %% of all wave-time"
          % ([int((e > span * f).sum()) for f in (0.5, 0.6, 0.7, 0.8, 0.9, 0.95)], live.sum(),
             100 * np.clip(e - np.median(e), 0, None).sum() / e.sum()))
    # a K2P_STATS build (HARE_LIB=...): what the slowest waves did after their tickets ran dry
    st3 = buf.cpu().numpy()[8 + 32:8 + 32 + 4 * W].reshape(W, 4)[:, 3]
    ph = buf.cpu().numpy()[8 + 32 + 4 * W:].reshape(W, 4).astype(np.float64) / 100.0
    if (st3 != 0).any():
        order = np.argsort(-end * live)[:12]
        for w in order:
            r, p_, c, e_ = (int(st3[w]) & 0xFFFF), (int(st3[w]) >> 16) & 0xFFFF, (int(st3[w]) >> 32) & 0xFFFF, (int(st3[w]) >> 48) & 0xFFFF
            print("   wave %4d: end %.0f us, %.0f us after dry: %d rounds (%.1f us each), %d pop steps, %d dense windows, %d exact phases" % (w, end[w], end[w] - dry[w], r, (end[w] - dry[w]) / max(r, 1), p_, c, e_))
            print("              rounds with <= 8 rays: %.0f us in pop steps, %.0f in dense windows, %.0f in the exact phase, %.0f in the rest of the round" % tuple(ph[w]))
        r = (st3[live] & 0xFFFF).astype(np.float64); d_ = (end - dry)[live]
        print("   all waves: %.1f rounds after dry on average, %.1f us each" % (r.mean(), d_.sum() / r.sum()))
