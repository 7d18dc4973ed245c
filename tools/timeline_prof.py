"""Developer tool: per-wave timeline of the production persistent voxel kernel (flag 0x2000): when each wave
started, took its last rays, and ended, grouped by residency tier (blockIdx // (grid/4))."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
os.environ["HARE_DEV"] = "1"      # the timeline bit is a developer flag; read when the scene is created
import hare_amd as H
N = int(os.environ.get("RAYS", 1 << 20)); D = 64
mesh = H.scenes.hall(); g = H.Voxel_Grid([H.Topology(mesh.verts, mesh.nverts)], D)
rays = H.scenes.burst_rays(N, mesh.size)
dr = torch.from_numpy(rays).cuda(); out = torch.empty(N * 56, dtype=torch.uint8, device="cuda")
W = 4096
buf = torch.zeros(8 + 32 + 4 * W, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for cfg in sys.argv[1:] or ["default"]:      # optional arguments: HARE_TICKET values to compare
    g.set_option("ticket_rays", 0 if cfg == "default" else int(cfg))
    for rep in range(2):
        buf.zero_()
        g.shoot_device(N, dr.data_ptr(), out.data_ptr(), d_counters=buf.data_ptr(), stream=st, flags=0x2000)
        torch.cuda.synchronize()
    raw = buf.cpu().numpy()[8 + 32:8 + 32 + 4 * W].reshape(W, 4).astype(np.uint64)
    tl = raw.astype(np.float64)
    live = tl[:, 0] > 0
    t0 = tl[live, 0].min()
    start = (tl[:, 0] - t0) / 100.0; last = (tl[:, 1] - t0) / 100.0; end = (tl[:, 2] - t0) / 100.0   # microseconds
    print("== %s: kernel span %.0f us, waves live %d" % (cfg, end[live].max(), live.sum()))
    if os.environ.get("HARE_VOXEL_KERNEL") == "pool":      # K1q: one workgroup per CU; slot 1 = tickets dry, slot 3 = rounds
        dry = last
        m = live
        # slot 3 (K1q): rounds | cooperative-tail rays << 16 | clock at the end of the pool rounds << 24 (40 bits)
        rounds = (raw[:, 3] & np.uint64(0xFFFF)).astype(np.float64); helped = ((raw[:, 3] >> np.uint64(16)) & np.uint64(0xFF)).astype(int)
        M40 = np.uint64((1 << 40) - 1)
        coop0 = (((raw[:, 3] >> np.uint64(24)) - (np.uint64(int(t0)) & M40)) & M40).astype(np.float64) / 100.0
        tl[:, 3] = rounds
        ct = end - coop0
        print("  pool rounds end p10/50/90/max %.0f/%.0f/%.0f/%.0f us; cooperative tail: rays per wave mean %.2f, duration mean %.1f p90 %.1f max %.1f us; "
              "of the latest 5 %% of waves: tail duration mean %.1f us, rays %.2f"
              % (*np.percentile(coop0[m], [10, 50, 90]), coop0[m].max(), helped[m].mean(), ct[m].mean(), np.percentile(ct[m], 90), ct[m].max(),
                 ct[m & (end > np.percentile(end[m], 95))].mean(), helped[m & (end > np.percentile(end[m], 95))].mean()))
        print("  wave starts p10/50/90/max %.1f/%.1f/%.1f/%.1f us after the first" % (*np.percentile(start[m], [10, 50, 90]), start[m].max()))
        print("  tickets dry p10/50/90 %.0f/%.0f/%.0f  end p10/50/90/max %.0f/%.0f/%.0f/%.0f  tail(end-dry) mean %.0f  rounds/wave mean %.0f"
              % (*np.percentile(dry[m], [10, 50, 90]), *np.percentile(end[m], [10, 50, 90]), end[m].max(), (end[m] - dry[m]).mean(), tl[m, 3].mean()))
        e = np.sort(end[live]); tot = e.max()
        print("  waves still running at 50/60/70/80/90/95%% of span: %s" % [int((e > tot * f).sum()) for f in (0.5, 0.6, 0.7, 0.8, 0.9, 0.95)])
        nw = int(os.environ.get("POOL_WAVES", 12))
        order = np.argsort(-end * live)[:12]
        print("  latest waves (block, wave, xcd=block%8, dry us, end us, rounds):", [(int(w // nw), int(w % nw), int((w // nw) % 8), int(dry[w]), int(end[w]), int(tl[w, 3])) for w in order])
        late = live & (end > np.percentile(end[live], 97))
        print("  the latest 3%%: blocks %s, rounds mean %.0f vs all %.0f, dry mean %.0f vs all %.0f" % (sorted(set(int(w // nw) for w in np.nonzero(late)[0]))[:40], tl[late, 3].mean(), tl[live, 3].mean(), dry[late].mean(), dry[live].mean()))
        continue
    tier = (np.arange(W) // 4) // 256
    for k in range(4):
        m = live & (tier == k)
        if not m.any():
            continue
        print("  tier %d: start %.0f..%.0f  last-refill p10/50/90 %.0f/%.0f/%.0f  end p10/50/90/max %.0f/%.0f/%.0f/%.0f  tail(end-last) mean %.0f"
              % (k, start[m].min(), start[m].max(), *np.percentile(last[m], [10, 50, 90]), *np.percentile(end[m], [10, 50, 90]), end[m].max(),
                 (end[m] - last[m]).mean()))
    e = np.sort(end[live]); tot = e.max()
    print("  waves still running at 50/60/70/80/90/95%% of span: %s" % [int((e > tot * f).sum()) for f in (0.5, 0.6, 0.7, 0.8, 0.9, 0.95)])
