"""Experiment: bounce rays (config 5) are incoherent -- how much would reordering them buy?
Generates third-bounce rays on the device, then times one cast as is / sorted by origin voxel /
sorted by origin voxel + direction octant / sorted by direction.  Kernel timing only; prints work counters too."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hare_amd as H
D = int(os.environ.get("DOMAIN", 128)); N = 1 << 20
scene = os.environ.get("SCENE", "cathedral")
mesh = H.scenes.SCENES[scene](); g = H.Voxel_Grid([H.Topology(mesh.verts, mesh.nverts)], D)
rays = H.scenes.burst_rays(N, mesh.size)
st = torch.cuda.current_stream().cuda_stream
d_rays = torch.from_numpy(rays).cuda(); d_out = torch.empty(N * 56, dtype=torch.uint8, device="cuda")
d_excl = torch.full((N,), -1, dtype=torch.int32, device="cuda")
for b in range(3):
    g.shoot_device(N, d_rays.data_ptr(), d_out.data_ptr(), flags=8, d_excl1=d_excl.data_ptr(), stream=st)
    g.reflect_device(N, d_rays.data_ptr(), d_out.data_ptr(), d_excl.data_ptr(), stream=st)
torch.cuda.synchronize()
r = d_rays.cpu().numpy().reshape(N, 6).copy(); e = d_excl.cpu().numpy().copy()
live = e != -2
print("third-bounce rays: %d live of %d" % (live.sum(), N))
def run(perm, label, K=10):
    rr = torch.from_numpy(np.ascontiguousarray(r[perm])).cuda(); ee = torch.from_numpy(np.ascontiguousarray(e[perm])).cuda()
    for _ in range(2): g.shoot_device(N, rr.data_ptr(), d_out.data_ptr(), d_excl1=ee.data_ptr(), stream=st)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K): g.shoot_device(N, rr.data_ptr(), d_out.data_ptr(), d_excl1=ee.data_ptr(), stream=st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / K
    print("%-44s %.3f ms  %.0f Mrays/s" % (label, ms, N / ms / 1e3), flush=True)
ident = np.arange(N)
run(ident, "as produced by the bounce loop")
ev, c = g.Shoot_batch(r, poly_origin1=e, count_work=True)
print("  work per ray: cells %.1f entries %.1f tests %.1f" % (c["cells"] / N, c["entries"] / N, c["tests"] / N))
i = g.info()
omin = np.array(i.obox_min); vd = np.array(i.voxel_dims)
cell = np.clip(np.floor((r[:, :3] - omin) / vd), 0, D - 1).astype(np.int64)
lin = (cell[:, 0] * D + cell[:, 1]) * D + cell[:, 2]
octant = ((r[:, 3] < 0).astype(np.int64) << 2) | ((r[:, 4] < 0).astype(np.int64) << 1) | (r[:, 5] < 0).astype(np.int64)
run(np.argsort(lin, kind="stable"), "sorted by origin voxel")
run(np.argsort(lin * 8 + octant, kind="stable"), "sorted by origin voxel, then direction octant")
run(np.argsort(octant * (D ** 3) + lin, kind="stable"), "sorted by direction octant, then origin voxel")
c8 = (cell // 8); lin8 = (c8[:, 0] * (D // 8) + c8[:, 1]) * (D // 8) + c8[:, 2]
run(np.argsort(lin8 * 8 + octant, kind="stable"), "sorted by 8^3 voxel block, then octant")
rng = np.random.default_rng(0)
run(rng.permutation(N), "random permutation")
