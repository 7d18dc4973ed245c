"""Developer statistic for the north_star's "triangle vertex tiles staged in LDS": in the production pool kernel (K1q), how many
DIFFERENT polygon records do the lanes of one cull batch fetch?  A wave-shared LDS tile can only save the difference
(lanes - distinct).  Burst in Fibonacci order and sorted by octahedral-Morton direction (the most coherent order there is)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["HARE_DEV"] = "1"
os.environ["HARE_VOXEL_KERNEL"] = "pool"
import numpy as np, torch
import hare_amd as H
from tools.coherence_exp import octa_key
N = 1 << 20
mesh = H.scenes.hall(); g = H.Voxel_Grid([H.Topology(mesh.verts, mesh.nverts)], 64)
rays = H.scenes.burst_rays(N, mesh.size)
st = torch.cuda.current_stream().cuda_stream
out = torch.empty(N * 56, dtype=torch.uint8, device="cuda")
for name, r in (("fibonacci order", rays), ("octahedral-Morton order (10 bits/axis)", rays[np.argsort(octa_key(rays[:, 3:], 10), kind="stable")])):
    dr = torch.from_numpy(np.ascontiguousarray(r)).cuda()
    buf = torch.zeros(8 + 32 + 4 * 4096, dtype=torch.int64, device="cuda")
    g.shoot_device(N, dr.data_ptr(), out.data_ptr(), d_counters=buf.data_ptr(), stream=st, flags=0x3000)
    torch.cuda.synchronize()
    lanes, distinct, batches = (int(x) for x in buf[8:11])
    print("%-42s cull batches %d, lanes/batch %.1f, distinct polygons/batch %.1f -> %.1f %% of the record fetches are repeats within the batch"
          % (name, batches, lanes / batches, distinct / batches, 100.0 * (1 - distinct / lanes)))
