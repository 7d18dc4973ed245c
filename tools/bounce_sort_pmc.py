"""Developer measurement (round 5): a LATE cast of the bounce loop with the live rays as the loop leaves them, or physically
re-ordered by (origin voxel block, direction) -- the order the VERDICT proposes -- for rocprofv3 --pmc (tools/bounce_sort_pmc.sh): K identical
casts of the front of bounce BOUNCE; the last K dispatches of the voxel kernel are the measurement.
    ORDER=given|sorted BOUNCE=5 RAYS=1048576 python tools/bounce_sort_pmc.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hare_amd as H
from tools.bounce_coherence_exp import morton3
from tools.coherence_exp import octa_key

N = int(os.environ.get("RAYS", 1 << 20)); KB = int(os.environ.get("BOUNCE", 5)); K = int(os.environ.get("K", 5)); order = os.environ.get("ORDER", "given")
mesh = H.scenes.cathedral(); g = H.Voxel_Grid([H.Topology(mesh.verts, mesh.nverts)], 128)
st = torch.cuda.current_stream().cuda_stream
d_rays = torch.from_numpy(H.scenes.burst_rays(N, mesh.size)).cuda(); d_out = torch.empty(N * 56, dtype=torch.uint8, device="cuda")
d_excl = torch.full((N,), -1, dtype=torch.int32, device="cuda")
for b in range(KB):
    g.shoot_device(N, d_rays.data_ptr(), d_out.data_ptr(), d_excl1=d_excl.data_ptr(), stream=st, flags=H.capi.SHOOT_RETIRED_RAYS)
    g.reflect_device(N, d_rays.data_ptr(), d_out.data_ptr(), d_excl.data_ptr(), stream=st)
torch.cuda.synchronize()
rays = d_rays.cpu().numpy().reshape(N, 6).copy(); excl = d_excl.cpu().numpy().copy()
perm = np.arange(N)
if order == "sorted":
    V = np.array(mesh.verts).reshape(-1, 3); lo, hi = V.min(0), V.max(0)
    o = np.nan_to_num((rays[:, :3] - lo) / (hi - lo)).clip(0, 1)
    c = (o * 31).astype(np.uint64)                       # 32^3 blocks of 4^3 voxels of the D = 128 grid
    d = np.nan_to_num(rays[:, 3:]); d[np.abs(d).sum(1) == 0] = (1, 0, 0)
    octant = ((d[:, 0] < 0).astype(np.uint64) << np.uint64(2)) | ((d[:, 1] < 0).astype(np.uint64) << np.uint64(1)) | (d[:, 2] < 0).astype(np.uint64)
    perm = np.argsort((morton3(c[:, 0], c[:, 1], c[:, 2], 5) << np.uint64(3)) | octant, kind="stable")
dr = torch.from_numpy(np.ascontiguousarray(rays[perm])).cuda(); de = torch.from_numpy(np.ascontiguousarray(excl[perm])).cuda()
for _ in range(2): g.shoot_device(N, dr.data_ptr(), d_out.data_ptr(), d_excl1=de.data_ptr(), stream=st, flags=H.capi.SHOOT_RETIRED_RAYS)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(K): g.shoot_device(N, dr.data_ptr(), d_out.data_ptr(), d_excl1=de.data_ptr(), stream=st, flags=H.capi.SHOOT_RETIRED_RAYS)
e1.record(); torch.cuda.synchronize()
print("order %s: front of bounce %d, %d live rays, %.4f ms per cast (%s)" % (order, KB, int((excl != -2).sum()), e0.elapsed_time(e1) / K, g.kernel_name(N)))
