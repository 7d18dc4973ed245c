"""Scratch GPU check: parity + rough kernel timing on the hall scene (not the bench)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import hare_amd as H
from oracle import pyoracle as po

D = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
t0 = time.time(); mesh = H.scenes.hall(); print("hall", mesh.P, "tris", time.time() - t0, flush=True)
T = H.Topology(mesh.verts, mesh.nverts)
t0 = time.time(); g = H.Voxel_Grid([T], D); print("build D=%d" % D, time.time() - t0, "items", g.info().total_items, flush=True)
rays = H.scenes.burst_rays(N, mesh.size)
print("runtime:", H.capi.lib.hare_hip_runtime_path())
# parity on a sample vs oracle
To = po.Topology(mesh.verts, mesh.nverts)
t0 = time.time(); go = po.VoxelGrid([To], domain=D); print("oracle build", time.time() - t0, flush=True)
so, io = go.lists(); s, i = g.Voxel_Inv(); print("lists equal:", np.array_equal(s, so) and np.array_equal(i, io))
ns = min(N, 200000)
idx = np.linspace(0, N - 1, ns).astype(np.int64)
t0 = time.time(); ref, c = go.shoot(rays[idx], nthreads=16); dt = time.time() - t0
print("oracle %d rays %.2fs (%.2f Mrays/s, 16 thr)" % (ns, dt, ns / dt / 1e6), c, flush=True)
ev, ctr = g.Shoot_batch(rays[idx], count_work=True)
print("gpu counters", ctr)
for f in ("hit", "poly_id", "t", "x", "y", "z"):
    print(f, "equal" if np.array_equal(ev[f], ref[f]) else "DIFF %d" % np.count_nonzero(ev[f] != ref[f]))
# device-resident timing
dr = torch.from_numpy(rays).cuda(); out = torch.empty(N * 56, dtype=torch.uint8, device="cuda")
ctr_d = torch.zeros(8, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3): g.shoot_device(N, dr.data_ptr(), out.data_ptr(), d_counters=ctr_d.data_ptr(), stream=st)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
K = 20
e0.record()
for _ in range(K): g.shoot_device(N, dr.data_ptr(), out.data_ptr(), d_counters=ctr_d.data_ptr(), stream=st)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / K
print("D=%d N=%d: %.3f ms/launch = %.1f Mrays/s" % (D, N, ms, N / ms / 1e3), "ctr", ctr_d.tolist()[:2])
e0.record()
for _ in range(K): g.shoot_device(N, dr.data_ptr(), out.data_ptr(), stream=st, flags=4)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / K
print("simple kernel: %.3f ms/launch = %.1f Mrays/s" % (ms, N / ms / 1e3))
