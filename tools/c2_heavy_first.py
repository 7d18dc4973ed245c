"""Developer experiment (round 5): does handing out the HEAVY rays first shorten a Voxel_Grid launch (K1q, C2)?

Two predictors of a ray's cost, all permuted on the host (class-major, the burst's own order inside a class), kernel time only,
events un-permuted and compared with the unpermuted run's (CRC):
  cells   VERDICT's estimate: voxels to the grid exit, sum_a |d_a| * t_exit / VoxelDims_a (set-up arithmetic of Voxel_Grid.cs:567-632)
  hit     PERFECT information about the walk: voxels to the HIT, sum_a |d_a| * t_hit / VoxelDims_a, t_hit from a previous cast --
          the bound on what any estimator of the walk length could deliver
Classes: the heaviest 12.5 / 25 / 50 % first, three classes (25 / 25 / 50), fully sorted, and a random permutation for scale.

    SCENE=hall DOMAIN=64 RAYS=1048576 python tools/c2_heavy_first.py
"""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hare_amd as H


def main():
    D = int(os.environ.get("DOMAIN", 64))
    mesh = H.scenes.SCENES[os.environ.get("SCENE", "hall")]()
    g = H.Voxel_Grid([H.Topology(mesh.verts, mesh.nverts)], D)
    st = torch.cuda.current_stream().cuda_stream
    V = np.asarray(mesh.verts).reshape(-1, 4, 3)[:, :3, :].reshape(-1, 3)
    lo, hi = V.min(0), V.max(0)
    vd = (hi - lo) / D
    for N in [int(x) for x in os.environ.get("RAYS", "1048576").split(",")]:
        rays = H.scenes.burst_rays(N, mesh.size)
        o, d = rays[:, :3], rays[:, 3:]
        with np.errstate(divide="ignore", invalid="ignore"):
            t0 = (lo - o) / d; t1 = (hi - o) / d
        t_exit = np.nanmin(np.maximum(t0, t1), 1)
        cells = (np.abs(d) * t_exit[:, None] / vd).sum(1)

        def run(perm, K=10):
            dr = torch.from_numpy(np.ascontiguousarray(rays[perm])).cuda(); out = torch.empty(N * 56, dtype=torch.uint8, device="cuda")
            for _ in range(3): g.shoot_device(N, dr.data_ptr(), out.data_ptr(), stream=st)
            torch.cuda.synchronize()
            best = 1e9
            for rep in range(3):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(K): g.shoot_device(N, dr.data_ptr(), out.data_ptr(), stream=st)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / K)
            ev = out.cpu().numpy().reshape(N, 56); back = np.empty_like(ev); back[perm] = ev
            return best, zlib.crc32(back.tobytes()), back

        t_base, c0, ev = run(np.arange(N))
        rec = np.frombuffer(ev.tobytes(), dtype=H.capi.XEVENT_DTYPE)
        t_hit = np.where(rec["hit"] != 0, rec["t"], t_exit)
        hitc = (np.abs(d) * t_hit[:, None] / vd).sum(1)
        print("%s D=%d n=%d kernel %s: as given %.4f ms = %.0f Mrays/s; cells-to-exit mean %.1f p99 %.1f; cells-to-hit mean %.1f p99 %.1f"
              % (mesh.name, D, N, g.kernel_name(N), t_base, N / t_base / 1e3, cells.mean(), np.percentile(cells, 99), hitc.mean(), np.percentile(hitc, 99)), flush=True)
        for name, p in (("cells-to-exit (VERDICT's estimate)", cells), ("cells-to-hit (perfect walk length)", hitc)):
            print("  predictor: %s" % name)
            for label, fr in (("heaviest 12.5 %% first", (0.125,)), ("heaviest 25 %% first", (0.25,)), ("heaviest 50 %% first", (0.5,)),
                              ("three classes 25 / 25 / 50", (0.25, 0.5))):
                thr = [np.quantile(p, 1 - f) for f in fr]
                cls = np.zeros(N, np.int64)
                for t in thr: cls += (p < t)
                perm = np.argsort(cls, kind="stable")
                t, c, _ = run(perm)
                print("     %-28s: %.4f ms (%+.1f %%)%s" % (label, t, 100 * (t / t_base - 1), "" if c == c0 else "  EVENTS DIFFER"), flush=True)
            # heavy rays first WITHIN each static chunk / ticket-sized run of 64 (locality of the burst kept: a permutation inside runs of 8192 rays)
            for run_len in (8192, 65536):
                perm = np.arange(N)
                for a in range(0, N, run_len):
                    seg = perm[a:a + run_len]
                    perm[a:a + run_len] = seg[np.argsort(-p[seg], kind="stable")]
                t, c, _ = run(perm)
                print("     %-28s: %.4f ms (%+.1f %%)%s" % ("sorted inside runs of %d" % run_len, t, 100 * (t / t_base - 1), "" if c == c0 else "  EVENTS DIFFER"), flush=True)
            perm = np.argsort(-p, kind="stable")
            t, c, _ = run(perm)
            print("     %-28s: %.4f ms (%+.1f %%)%s" % ("fully sorted, descending", t, 100 * (t / t_base - 1), "" if c == c0 else "  EVENTS DIFFER"), flush=True)
        # the LAST part of the batch first (the tickets serve the burst's end, then its start): does the order of the bands matter at all?
        perm = np.concatenate([np.arange(N // 2, N), np.arange(0, N // 2)])
        t, c, _ = run(perm)
        print("  %-31s: %.4f ms (%+.1f %%)%s" % ("second half of the burst first", t, 100 * (t / t_base - 1), "" if c == c0 else "  EVENTS DIFFER"))
        t, c, _ = run(np.random.default_rng(0).permutation(N))
        print("  %-31s: %.4f ms (%+.1f %%)%s" % ("random permutation", t, 100 * (t / t_base - 1), "" if c == c0 else "  EVENTS DIFFER"), flush=True)


if __name__ == "__main__":
    main()
