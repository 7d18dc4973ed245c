"""Developer experiment: how much would a PHYSICALLY sorted ray front be worth in the bounce loop (config 5)?
Takes the ray front of bounce K (device-resident shoot -> reflect, as bench.py does), pulls it to the host, permutes rays and
their poly_origin array together by several keys, and times the shoot kernel on each order.  Kernel time only; the events of
every order are compared (un-permuted) with the unsorted run's: the order must not change any X_Event.

    SCENE=cathedral DOMAIN=128 RAYS=1048576 BOUNCE=4 python tools/bounce_coherence_exp.py
"""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools.coherence_exp import octa_key


def morton3(a, b, c, bits):
    k = np.zeros(a.shape, np.uint64)
    for i in range(bits):
        for j, x in enumerate((a, b, c)):
            k |= ((x >> np.uint64(i)) & np.uint64(1)) << np.uint64(3 * i + j)
    return k


def main():
    import torch
    import hare_amd as H
    D = int(os.environ.get("DOMAIN", 128)); N = int(os.environ.get("RAYS", 1 << 20)); KB = int(os.environ.get("BOUNCE", 4))
    mesh = H.scenes.SCENES[os.environ.get("SCENE", "cathedral")](); g = H.Voxel_Grid([H.Topology(mesh.verts, mesh.nverts)], D)
    rays0 = H.scenes.burst_rays(N, mesh.size)
    st = torch.cuda.current_stream().cuda_stream
    d_rays = torch.from_numpy(rays0).cuda(); d_out = torch.empty(N * 56, dtype=torch.uint8, device="cuda")
    d_excl = torch.full((N,), -1, dtype=torch.int32, device="cuda")
    for b in range(KB):                                   # advance to the front of bounce KB
        g.shoot_device(N, d_rays.data_ptr(), d_out.data_ptr(), d_excl1=d_excl.data_ptr(), stream=st, flags=H.capi.SHOOT_RETIRED_RAYS)
        g.reflect_device(N, d_rays.data_ptr(), d_out.data_ptr(), d_excl.data_ptr(), stream=st)
    torch.cuda.synchronize()
    rays = d_rays.cpu().numpy().reshape(N, 6).copy(); excl = d_excl.cpu().numpy().copy()
    live = excl != -2
    print("kernel %s, front of bounce %d: %d rays, %d live" % (g.kernel_name(N), KB, N, int(live.sum())))

    def run(perm, K=10):
        r = np.ascontiguousarray(rays[perm]); e = np.ascontiguousarray(excl[perm])
        dr = torch.from_numpy(r).cuda(); de = torch.from_numpy(e).cuda(); out = torch.empty(N * 56, dtype=torch.uint8, device="cuda")
        for _ in range(2): g.shoot_device(N, dr.data_ptr(), out.data_ptr(), d_excl1=de.data_ptr(), stream=st, flags=H.capi.SHOOT_RETIRED_RAYS)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(K): g.shoot_device(N, dr.data_ptr(), out.data_ptr(), d_excl1=de.data_ptr(), stream=st, flags=H.capi.SHOOT_RETIRED_RAYS)
        e1.record(); torch.cuda.synchronize()
        ev = out.cpu().numpy().reshape(N, 56)
        back = np.empty_like(ev); back[perm] = ev         # events in the original ray order
        return e0.elapsed_time(e1) / K, zlib.crc32(back.tobytes())

    ident = np.arange(N)
    t0, crc0 = run(ident)
    print("  as the loop leaves them        : %.3f ms" % t0)
    lo = np.array(mesh.verts).reshape(-1, 3).min(0); hi = np.array(mesh.verts).reshape(-1, 3).max(0)
    o = np.nan_to_num((rays[:, :3] - lo) / (hi - lo)).clip(0, 1)
    orders = {}
    for bits in (4, 6, 8):
        q = (1 << bits) - 1
        c = (o * q).astype(np.uint64)
        orders["origin morton %d bits/axis" % bits] = morton3(c[:, 0], c[:, 1], c[:, 2], bits)
    d = np.nan_to_num(rays[:, 3:]); d[np.abs(d).sum(1) == 0] = (1, 0, 0)
    orders["direction octa-morton 6 bits"] = octa_key(d, 6)
    c = (o * 63).astype(np.uint64)
    orders["origin 6 bits, then direction 3"] = (morton3(c[:, 0], c[:, 1], c[:, 2], 6) << np.uint64(6)) | octa_key(d, 3)
    c = (o * 15).astype(np.uint64)
    orders["direction 4 bits, then origin 4"] = (octa_key(d, 4) << np.uint64(12)) | morton3(c[:, 0], c[:, 1], c[:, 2], 4)
    for name, key in orders.items():
        perm = np.argsort(key, kind="stable")
        t, crc = run(perm)
        print("  %-31s: %.3f ms (%+.1f %%)%s" % (name, t, 100 * (t / t0 - 1), "" if crc == crc0 else "  EVENTS DIFFER"))
    t, crc = run(np.random.default_rng(0).permutation(N))
    print("  %-31s: %.3f ms (%+.1f %%)%s" % ("random permutation", t, 100 * (t / t0 - 1), "" if crc == crc0 else "  EVENTS DIFFER"))


if __name__ == "__main__":
    main()
