"""Experiment the north_star names: does re-ordering the burst by direction (octahedral-Morton key, the most coherent order
there is) help?  Host-side sort, kernel timing only, for whichever voxel kernel HARE_VOXEL_KERNEL selects."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def morton2(a, b, bits):
    k = np.zeros(a.shape, np.uint64)
    for i in range(bits):
        k |= ((a >> np.uint64(i)) & np.uint64(1)) << np.uint64(2 * i) | ((b >> np.uint64(i)) & np.uint64(1)) << np.uint64(2 * i + 1)
    return k


def octa_key(d, bits):
    n = d / np.abs(d).sum(1, keepdims=True)
    u, v = n[:, 0].copy(), n[:, 1].copy()
    neg = n[:, 2] < 0
    uu = (1 - np.abs(v)) * np.sign(u); vv = (1 - np.abs(u)) * np.sign(v)
    u[neg], v[neg] = uu[neg], vv[neg]
    q = (1 << bits) - 1
    a = np.clip(((u * 0.5 + 0.5) * q), 0, q).astype(np.uint64); b = np.clip(((v * 0.5 + 0.5) * q), 0, q).astype(np.uint64)
    return morton2(a, b, bits)


def main():
    import torch
    import hare_amd as H
    D = int(os.environ.get("DOMAIN", 64)); N = int(os.environ.get("RAYS", 1 << 20))
    mesh = H.scenes.SCENES[os.environ.get("SCENE", "hall")](); g = H.Voxel_Grid([H.Topology(mesh.verts, mesh.nverts)], D)
    rays = H.scenes.burst_rays(int(os.environ.get("BURST", N)), mesh.size, start=int(os.environ.get("START", 0)), count=N)

    def run(r, K=20):
        dr = torch.from_numpy(np.ascontiguousarray(r)).cuda(); out = torch.empty(N * 56, dtype=torch.uint8, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3): g.shoot_device(N, dr.data_ptr(), out.data_ptr(), stream=st)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(K): g.shoot_device(N, dr.data_ptr(), out.data_ptr(), stream=st)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / K

    print("kernel %s, %d rays" % (g.kernel_name(N), N))
    print("  fibonacci order               : %.3f ms" % run(rays))
    for bits in (4, 6, 8, 10):
        perm = np.argsort(octa_key(rays[:, 3:], bits), kind="stable")
        print("  octahedral morton %2d bits/axis: %.3f ms" % (bits, run(rays[perm])))
    rng = np.random.default_rng(0)
    print("  random permutation            : %.3f ms" % run(rays[rng.permutation(N)]))


if __name__ == "__main__":
    main()
