"""PCIe-inclusive rate of the host-buffer entry point hare_shoot_batch: pageable vs pinned caller buffers."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hare_amd as H
from hare_amd import capi
mesh = H.scenes.hall(); g = H.Voxel_Grid([H.Topology(mesh.verts, mesh.nverts)], 64)
N = 1 << 20
rays = H.scenes.burst_rays(N, mesh.size)
def rate(r, out, label):
    ctr = capi.Counters()
    def call(): capi.check(capi.lib.hare_shoot_batch(g._h, 0, 0, N, r.ctypes.data, None, None, 0, out.ctypes.data, C.addressof(ctr)))
    call(); ts = []
    for _ in range(7):
        t0 = time.perf_counter(); call(); ts.append(time.perf_counter() - t0)
    print("%-34s best %.2f ms = %6.1f Mrays/s (hits %d)" % (label, min(ts) * 1e3, N / min(ts) / 1e6, ctr.hits))
rate(rays, np.zeros(N, capi.XEVENT_DTYPE), "pageable numpy arrays")
pr = torch.from_numpy(rays).pin_memory(); po_ = torch.zeros(N * 56, dtype=torch.uint8).pin_memory()
rate(pr.numpy(), po_.numpy().view(capi.XEVENT_DTYPE), "pinned (torch pin_memory) arrays")
t0 = time.perf_counter(); x = torch.from_numpy(rays).pin_memory(); print("pin_memory of 50 MB: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
