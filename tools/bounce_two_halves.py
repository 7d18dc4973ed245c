"""Developer measurement: the device-resident bounce loop (config 5) as ONE batch on one stream against the same rays as TWO (or more) independent
part-batches, each on its own stream (shoot -> reflect -> shoot ... per part): one part's end of launch runs under the other's
steady state.  Rays are independent, so the events are the same bytes.  usage: [SCENE=cathedral DOMAIN=128 RAYS=1048576 B=8] python tools/bounce_two_halves.py [parts ...]"""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hare_amd as H
N = int(os.environ.get("RAYS", 1 << 20)); D = int(os.environ.get("DOMAIN", 128)); B = int(os.environ.get("B", 8))
mesh = getattr(H.scenes, os.environ.get("SCENE", "cathedral"))()
g = H.Voxel_Grid([H.Topology(mesh.verts, mesh.nverts)], D)
rays0 = torch.from_numpy(H.scenes.burst_rays(N, mesh.size)).cuda()
ref = None
for parts in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]:
    bounds = [N * k // parts for k in range(parts + 1)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    rays = rays0.clone(); out = torch.empty(N * 56, dtype=torch.uint8, device="cuda"); excl = torch.empty(N, dtype=torch.int32, device="cuda")
    last = torch.empty(N * 56, dtype=torch.uint8, device="cuda")
    def loop():
        rays.copy_(rays0); excl.fill_(-1)
        e = torch.cuda.Event(); e.record(torch.cuda.current_stream())
        for k, s in enumerate(streams):
            s.wait_event(e)
        for b in range(B):
            for k, s in enumerate(streams):
                lo, n = bounds[k], bounds[k + 1] - bounds[k]
                rp, op, ep = rays.data_ptr() + lo * 48, out.data_ptr() + lo * 56, excl.data_ptr() + lo * 4
                g.shoot_device(n, rp, op, d_excl1=ep, stream=s.cuda_stream, flags=H.capi.SHOOT_RETIRED_RAYS)
                if b + 1 < B:
                    g.reflect_device(n, rp, op, ep, stream=s.cuda_stream)
        for s in streams:
            e2 = torch.cuda.Event(); e2.record(s); torch.cuda.current_stream().wait_event(e2)
    loop(); torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); loop(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    crc = zlib.crc32(out.cpu().numpy().tobytes())
    if ref is None: ref = crc
    print("%d part(s) of %d rays, kernel %s: %.3f ms per %d-cast loop = %.0f Mcasts/s (upper bound: retired rays count); last cast's events identical: %s"
          % (parts, bounds[1], g.kernel_name(bounds[1]), best, B, N * B / best / 1e3, crc == ref), flush=True)
