"""Developer tool: where a LATE cast of the bounce loop (config 5) spends its time.  Runs CASTS-1 casts + reflections of a 1M-ray burst
in the cathedral (D = 128), then the next cast under the timeline flag (0x2000): per wave {start, tickets dry, end, rounds} on the
100 MHz clock.  Optional arguments: ticket_rays values to compare (0 = the library's rule).
env: CASTS (default 6), RAYS, KERNEL (pool | persist)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["HARE_DEV"] = "1"
import numpy as np, torch
import hare_amd as H
from hare_amd import capi

N = int(os.environ.get("RAYS", 1 << 20)); CASTS = int(os.environ.get("CASTS", 6)); kern = os.environ.get("KERNEL", "pool")
mesh = H.scenes.cathedral(); g = H.Voxel_Grid([H.Topology(mesh.verts, mesh.nverts)], 128)
g.set_option("voxel_kernel", 2 if kern == "pool" else 1)
rays = H.scenes.burst_rays(N, mesh.size)
st = torch.cuda.current_stream().cuda_stream
d_rays = torch.from_numpy(rays).cuda(); d_ev = torch.empty(N * 56, dtype=torch.uint8, device="cuda")
d_ex = torch.full((N,), -1, dtype=torch.int32, device="cuda")
for b in range(CASTS - 1):
    g.shoot_device(N, d_rays.data_ptr(), d_ev.data_ptr(), d_excl1=d_ex.data_ptr(), stream=st, flags=capi.SHOOT_RETIRED_RAYS)
    g.reflect_device(N, d_rays.data_ptr(), d_ev.data_ptr(), d_ex.data_ptr(), stream=st)
torch.cuda.synchronize()
WPB = 12 if kern == "pool" else 4
W = 256 * 16
buf = torch.zeros(8 + 32 + 4 * W, dtype=torch.int64, device="cuda")
for cfg in sys.argv[1:] or ["0"]:
    g.set_option("ticket_rays", int(cfg))
    for rep in range(2):
        buf.zero_()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        g.shoot_device(N, d_rays.data_ptr(), d_ev.data_ptr(), d_excl1=d_ex.data_ptr(), d_counters=buf.data_ptr(), stream=st,
                       flags=capi.SHOOT_RETIRED_RAYS | 0x2000)
        e1.record(); torch.cuda.synchronize()
    raw = buf.cpu().numpy()[8 + 32:].reshape(W, 4)
    tl = raw.astype(np.float64)
    if kern == "pool":       # K1q packs slot 3: rounds | cooperative-tail rays << 16 | clock at the end of the pool rounds << 24
        tl[:, 3] = (raw[:, 3].astype(np.uint64) & np.uint64(0xFFFF)).astype(np.float64)
    live = tl[:, 0] > 0
    t0 = tl[live, 0].min()
    start = (tl[:, 0] - t0) / 100.0; dry = (tl[:, 1] - t0) / 100.0; end = (tl[:, 2] - t0) / 100.0
    print("== cast %d, %s kernel, ticket_rays %s: %.3f ms (event), span %.0f us, waves %d" % (CASTS, kern, cfg, e0.elapsed_time(e1), end[live].max(), live.sum()))
    m = live
    if kern == "pool":
        print("  tickets dry p10/50/90 %.0f/%.0f/%.0f us | end p10/50/90/max %.0f/%.0f/%.0f/%.0f | tail (end - dry) mean %.0f p90 %.0f | rounds/wave mean %.0f max %.0f"
              % (*np.percentile(dry[m], [10, 50, 90]), *np.percentile(end[m], [10, 50, 90]), end[m].max(), (end[m] - dry[m]).mean(),
                 np.percentile(end[m] - dry[m], 90), tl[m, 3].mean(), tl[m, 3].max()))
    else:
        print("  last refill p10/50/90 %.0f/%.0f/%.0f us | end p10/50/90/max %.0f/%.0f/%.0f/%.0f" % (*np.percentile(dry[m], [10, 50, 90]), *np.percentile(end[m], [10, 50, 90]), end[m].max()))
    e = np.sort(end[m]); tot = e.max()
    print("  waves still running at 50/60/70/80/90/95 %% of the span: %s of %d;  mean wave lifetime / span = %.2f"
          % ([int((e > tot * f).sum()) for f in (0.5, 0.6, 0.7, 0.8, 0.9, 0.95)], m.sum(), (end[m] - start[m]).mean() / tot))
