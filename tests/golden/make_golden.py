"""Generates tests/golden/c1_shoebox.npz -- BASELINE.json config 1 (10k random rays into the 1k-tri
shoebox) plus a block of edge-case rays, with the X_Event records the ORACLE returns for the voxel,
octree and kd-tree paths.

The reference has no fixtures and cannot be run (C#, no .NET toolchain): these vectors pin the
oracle against regressions and give the GPU tests a committed expected output; they are NOT
outputs of the reference itself (parity unpinned, see oracle/hare_oracle.h).

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import hare_amd.scenes as scenes  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

VOXEL_D = 8
OCT = (5, 8)
KD = (8, 8)


def edge_rays(size):
    L = np.asarray(size)
    c = 0.5 * L
    rows = []
    for a in range(3):                      # axis-aligned both ways from the centre and from outside
        for s in (1.0, -1.0):
            d = np.zeros(3)
            d[a] = s
            rows.append(np.concatenate([c, d]))
            o = c.copy()
            o[a] = -3.0 if s > 0 else L[a] + 3.0
            rows.append(np.concatenate([o, d]))          # starts outside, points in
            rows.append(np.concatenate([o, -d]))         # starts outside, points away
    for d in ([1, 1, 0], [1, 0, 1], [0, 1, 1], [1, 1, 1], [-1, 1, -1], [1, -1, 1e-17], [-0.0, 0.0, 1.0],
              [0.0, -0.0, -1.0], [2.0, 0.0, 0.0], [1e-3, 1e-3, 1.0]):
        rows.append(np.concatenate([c, np.asarray(d, float)]))
    rows.append(np.concatenate([c, [0.0, 0.0, 0.0]]))     # zero direction
    rows.append(np.array([1e6, 1e6, 1e6, -1.0, -1.0, -1.0]))
    rows.append(np.array([np.nan, 1.0, 1.0, 1.0, 0.0, 0.0]))
    rows.append(np.array([1.0, 1.0, 1.0, np.nan, 0.0, 1.0]))
    rows.append(np.array([1.0, 1.0, 1.0, np.inf, 0.0, 0.0]))
    # corners / edges of the box
    rows.append(np.concatenate([c, L - c]))                # towards the (Lx,Ly,Lz) corner
    rows.append(np.concatenate([c, -c]))                   # towards the origin corner
    return np.array(rows)


def main():
    mesh = scenes.shoebox()
    T = po.Topology(mesh.verts, mesh.nverts)
    rays = np.concatenate([scenes.random_rays(10000, mesh.size), edge_rays(mesh.size)])
    n = rays.shape[0]
    vox = po.VoxelGrid([T], domain=VOXEL_D, build_mode=0)
    ev_v, ctr_v = vox.shoot(rays)
    # exclusion overload: re-shoot every ray excluding the polygon it first hit
    e1 = ev_v["poly_id"].astype(np.int32)
    ev_vx, _ = vox.shoot(rays, excl1=e1)
    oc = po.Octree([T], *OCT)
    ev_o, ctr_o = oc.shoot(rays)
    ev_ox, _ = oc.shoot(rays, excl1=e1)
    kd = po.KDTree([T], *KD)
    ev_k, ctr_k = kd.shoot(rays)
    out = os.path.join(ROOT, "tests", "golden", "c1_shoebox.npz")
    np.savez_compressed(out, rays=rays, excl1=e1, voxel=ev_v, voxel_excl=ev_vx, octree=ev_o, octree_excl=ev_ox,
                        kdtree=ev_k, voxel_ctr=np.array([ctr_v[k] for k in ("cells", "entries", "tests")]),
                        octree_ctr=np.array([ctr_o[k] for k in ("cells", "entries", "tests")]),
                        params=np.array([VOXEL_D, OCT[0], OCT[1], KD[0], KD[1]]))
    print("wrote", out, n, "rays;", int(ev_v["hit"].sum()), "voxel hits;", os.path.getsize(out) // 1024, "KiB")


if __name__ == "__main__":
    main()
