"""Developer experiment: the octree kernel on a burst physically sorted by direction (octahedral-Morton key), host-side sort,
kernel time only; events un-permuted and compared."""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hare_amd as H
from tools.coherence_exp import octa_key
mesh = H.scenes.SCENES[os.environ.get("SCENE", "hall")]()
g = H.Octree([H.Topology(mesh.verts, mesh.nverts)], 8, 16)
st = torch.cuda.current_stream().cuda_stream
for N in [int(x) for x in os.environ.get("RAYS", "1048576").split(",")]:
    rays = H.scenes.burst_rays(N, mesh.size)

    def run(perm, K=6):
        dr = torch.from_numpy(np.ascontiguousarray(rays[perm])).cuda(); out = torch.empty(N * 56, dtype=torch.uint8, device="cuda")
        for _ in range(2): g.shoot_device(N, dr.data_ptr(), out.data_ptr(), stream=st)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(K): g.shoot_device(N, dr.data_ptr(), out.data_ptr(), stream=st)
        e1.record(); torch.cuda.synchronize()
        ev = out.cpu().numpy().reshape(N, 56); back = np.empty_like(ev); back[perm] = ev
        return e0.elapsed_time(e1) / K, zlib.crc32(back.tobytes())
    t0, c0 = run(np.arange(N))
    print("octree %s n=%d: fibonacci order %.3f ms" % (mesh.name, N, t0))
    for bits in (3, 4, 6, 8):
        t, c = run(np.argsort(octa_key(rays[:, 3:], bits), kind="stable"))
        print("   octahedral morton %d bits/axis: %.3f ms (%+.1f %%)%s" % (bits, t, 100 * (t / t0 - 1), "" if c == c0 else "  EVENTS DIFFER"))
    z = rays[:, 5] / np.linalg.norm(rays[:, 3:], axis=1)
    az = np.arctan2(rays[:, 4], rays[:, 3])
    for nb in (16, 64):
        key = (np.clip(((z + 1) / 2 * nb).astype(int), 0, nb - 1) * 4096 + ((az + np.pi) / (2 * np.pi) * 4095).astype(int))
        t, c = run(np.argsort(key, kind="stable"))
        print("   %d polar bands, azimuth inside a band: %.3f ms (%+.1f %%)%s" % (nb, t, 100 * (t / t0 - 1), "" if c == c0 else "  EVENTS DIFFER"))
