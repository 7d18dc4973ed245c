"""Developer experiment: does starting the HEAVY rays first shorten the octree launch's drain (tools/timeline_oct.py)?
A cheap predictor of a ray's work -- the line integral of a 16^3 grid of polygon counts along the ray's chord through the scene
box (64 samples) -- picks the heaviest rays; the batch is permuted on the host (heavy classes first, original order inside a
class) and K2p is timed on it.  Events are un-permuted and compared with the unpermuted run's."""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hare_amd as H


def predictor(verts, rays, G=16, S=64):
    V = np.asarray(verts).reshape(-1, 4, 3)[:, :3, :]
    lo = V.reshape(-1, 3).min(0); hi = V.reshape(-1, 3).max(0)
    cen = V.mean(1)
    cell = np.clip(((cen - lo) / (hi - lo) * G).astype(int), 0, G - 1)
    dens = np.zeros((G, G, G)); np.add.at(dens, (cell[:, 0], cell[:, 1], cell[:, 2]), 1)
    out = np.empty(len(rays))
    for a in range(0, len(rays), 1 << 16):
        r = rays[a:a + (1 << 16)]; o = r[:, :3]; d = r[:, 3:]
        with np.errstate(divide="ignore", invalid="ignore"):
            t0 = (lo - o) / d; t1 = (hi - o) / d
        tn = np.clip(np.nanmax(np.minimum(t0, t1), 1), 0, None); tf = np.nanmin(np.maximum(t0, t1), 1)
        ln = np.clip(tf - tn, 0, None)
        ts = tn[:, None] + ln[:, None] * (np.arange(S) + 0.5) / S
        pts = o[:, None, :] + d[:, None, :] * ts[:, :, None]
        c = np.clip(np.nan_to_num((pts - lo) / (hi - lo) * G).astype(int), 0, G - 1)
        out[a:a + len(r)] = dens[c[:, :, 0], c[:, :, 1], c[:, :, 2]].sum(1) * ln / S
    return out


def main():
    kind = os.environ.get("KIND", "octree")
    mesh = H.scenes.SCENES[os.environ.get("SCENE", "hall")]()
    T = H.Topology(mesh.verts, mesh.nverts)
    g = H.Octree([T], 8, 16) if kind == "octree" else H.Voxel_Grid([T], int(os.environ.get("DOMAIN", 64)))
    st = torch.cuda.current_stream().cuda_stream
    for N in [int(x) for x in os.environ.get("RAYS", "1048576").split(",")]:
        rays = H.scenes.burst_rays(N, mesh.size)
        p = predictor(mesh.verts, rays)

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
        print("%s %s n=%d kernel %s: as given %.3f ms" % (kind, mesh.name, N, g.kernel_name(N), t0))
        for frac in (0.05, 0.125, 0.25, 0.5):
            thr = np.quantile(p, 1 - frac)
            perm = np.concatenate([np.nonzero(p >= thr)[0], np.nonzero(p < thr)[0]])
            t, c = run(perm)
            print("   heaviest %4.1f %% (by the predictor) first: %.3f ms (%+.1f %%)%s" % (100 * frac, t, 100 * (t / t0 - 1), "" if c == c0 else "  EVENTS DIFFER"))
        perm = np.argsort(-p, kind="stable")
        t, c = run(perm)
        print("   fully sorted by the predictor, descending: %.3f ms (%+.1f %%)%s" % (t, 100 * (t / t0 - 1), "" if c == c0 else "  EVENTS DIFFER"))
        key = np.minimum(15, (4 * p / p.mean()).astype(int))
        perm = np.argsort(-key, kind="stable")
        t, c = run(perm)
        print("   16 classes of 4p/mean, heaviest class first: %.3f ms (%+.1f %%)%s" % (t, 100 * (t / t0 - 1), "" if c == c0 else "  EVENTS DIFFER"))
        t, c = run(np.random.default_rng(0).permutation(N))
        print("   random permutation: %.3f ms (%+.1f %%)%s" % (t, 100 * (t / t0 - 1), "" if c == c0 else "  EVENTS DIFFER"))


if __name__ == "__main__":
    main()
