"""Measures the PCIe-inclusive rate of the host-buffer entry point hare_shoot_batch (DESIGN.md note)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hare_amd as H
mesh = H.scenes.hall(); g = H.Voxel_Grid([H.Topology(mesh.verts, mesh.nverts)], 64)
N = 1 << 20
rays = H.scenes.burst_rays(N, mesh.size)
g.Shoot_batch(rays)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); g.Shoot_batch(rays); ts.append(time.perf_counter() - t0)
print("hare_shoot_batch (pageable host buffers, H2D + kernel + D2H): best %.2f ms = %.1f Mrays/s" % (min(ts) * 1e3, N / min(ts) / 1e6))
