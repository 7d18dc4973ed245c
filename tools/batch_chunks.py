"""Developer measurement: hare_shoot_batch from host buffers (H2D + kernel + D2H) over the number of pipelined chunks (scene option
"batch_chunks"), full X_Event records and 16-byte slim records, 1M and 4M rays."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hare_amd as H
from hare_amd import capi
mesh = H.scenes.hall(); g = H.Voxel_Grid([H.Topology(mesh.verts, mesh.nverts)], 64)
for N in (1 << 20, 1 << 22):
    rays = H.scenes.burst_rays(N, mesh.size)
    out = np.zeros(N, capi.XEVENT_DTYPE); slim = np.zeros(N * 16, np.uint8)
    for K in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 6, 8]:
        g.set_option("batch_chunks", K)
        res = []
        for flags, buf in ((0, out), (capi.SHOOT_SLIM_EVENTS, slim)):
            ctr = capi.Counters()
            def call(): capi.check(capi.lib.hare_shoot_batch(g._h, 0, 0, N, rays.ctypes.data, None, None, flags, buf.ctypes.data, C.addressof(ctr)))
            call(); ts = []
            for _ in range(7):
                t0 = time.perf_counter(); call(); ts.append(time.perf_counter() - t0)
            res.append(N / min(ts) / 1e6)
        print("n=%d chunks=%d: full records %.0f Mrays/s, slim records %.0f Mrays/s" % (N, K, res[0], res[1]), flush=True)
