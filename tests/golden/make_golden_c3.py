"""Generates tests/golden/c3_quads.npz -- the third committed vector set: QUADRILATERALS (Hare_Geometry_Polygons.cs:731-823), which the
first two sets do not contain.

  * a 6 x 5 x 4 m room whose six faces are 5 x 5 lattices of rectangles (150 quadrilaterals), plus 260 quadrilaterals inside it --
    general convex ones in axis-aligned planes (four lattice corners, not parallelograms), tilted parallelograms (exactly planar on the
    lattice), a few coincident twins -- plus 120 triangles, interleaved; every coordinate on the 2^-8 m lattice;
  * rays from seeded random origins aimed exactly at corners, edge midpoints, the midpoint of the diagonal v0 - v2 BOTH triangles of
    Quadrilateral.Intersect share, and interior points.

Expected X_Events from the ORACLE (not from the reference, which cannot be run here: parity unpinned): Voxel_Grid D = 8 plain and with
poly_origin1 = the polygon first hit, Octree 5 / 6 plain and with the exclusion, KDTree 7 / 6.  bindings/csharp/tests/GoldenParity.cs
(case 3) runs the reference classes on the same inputs.

Run from the repo root:  python tests/golden/make_golden_c3.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import hare_amd.scenes as scenes  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

VOXEL_D = 8
OCT = (5, 6)
KD = (7, 6)
SIZE = (6.0, 5.0, 4.0)


def quad_scene(seed=23):
    rng = np.random.default_rng(seed)
    L = np.asarray(SIZE)
    polys = []
    scenes._QUADS = True
    try:
        n = 5
        faces = [([0, 0, 0], [L[0], 0, 0], [0, L[1], 0]), ([0, 0, L[2]], [L[0], 0, 0], [0, L[1], 0]), ([0, 0, 0], [0, L[1], 0], [0, 0, L[2]]),
                 ([L[0], 0, 0], [0, L[1], 0], [0, 0, L[2]]), ([0, 0, 0], [L[0], 0, 0], [0, 0, L[2]]), ([0, L[1], 0], [L[0], 0, 0], [0, 0, L[2]])]
        room = np.concatenate([scenes._patch(o, u, v, n, n) for o, u, v in faces], axis=0)        # [150, 4, 3]
    finally:
        scenes._QUADS = False
    for q in room:
        polys.append((q, 4))
    # general convex quadrilaterals in axis-aligned planes: corners of a jittered rectangle, all on the lattice, one coordinate constant
    for k in range(160):
        axis = k % 3
        c = rng.uniform(0.8, 0.2 + L.min() - 1.0, 3) * (L / L.min()) * 0.8 + 0.3
        w, h = rng.uniform(0.3, 0.9, 2)
        uv = np.array([[0, 0], [w, 0], [w, h], [0, h]]) + rng.uniform(-0.12, 0.12, (4, 2))
        q = np.tile(c, (4, 1))
        a, b = [(1, 2), (0, 2), (0, 1)][axis]
        q[:, a] += uv[:, 0]; q[:, b] += uv[:, 1]
        polys.append((scenes.snap(q), 4))
    # tilted parallelograms: c, c + a, c + a + b, c + b with lattice a, b (exactly planar)
    for k in range(80):
        c = scenes.snap(rng.uniform(0.6, 3.0, 3))
        a = scenes.snap(rng.uniform(-0.7, 0.7, 3)); b = scenes.snap(rng.uniform(-0.7, 0.7, 3))
        if np.linalg.norm(np.cross(a, b)) < 0.02:
            b = b + scenes.snap([0.25, -0.125, 0.375])
        polys.append((np.array([c, c + a, c + a + b, c + b]), 4))
    for k in range(120):
        c = scenes.snap(rng.uniform(0.5, 3.5, 3))
        a = scenes.snap(rng.uniform(-0.6, 0.6, 3)); b = scenes.snap(rng.uniform(-0.6, 0.6, 3))
        if np.linalg.norm(np.cross(a, b)) < 0.02:
            b = b + scenes.snap([0.125, 0.25, -0.375])
        t = np.zeros((4, 3)); t[:3] = [c, c + a, c + b]
        polys.append((t, 3))
    for k in range(20):                                     # coincident twins of earlier quadrilaterals (list order decides a tie)
        polys.append((polys[150 + 7 * k][0].copy(), 4))
    order = np.concatenate([np.arange(150), 150 + rng.permutation(len(polys) - 150)])          # the room first (its min corner is the origin)
    verts = np.zeros((len(polys), 4, 3)); nverts = np.zeros(len(polys), np.int32)
    for i, j in enumerate(order):
        verts[i], nverts[i] = polys[j][0], polys[j][1]
    verts = np.clip(verts, 0.0, L)                          # stay inside the room (clipping keeps lattice values)
    # drop degenerate polygons a clip could have made
    e1 = verts[:, 1] - verts[:, 0]; e2 = verts[:, 2] - verts[:, 0]
    keep = np.linalg.norm(np.cross(e1, e2), axis=1) > 1e-6
    e3 = verts[:, 3] - verts[:, 0]
    keep &= (nverts == 3) | (np.linalg.norm(np.cross(e2, e3), axis=1) > 1e-6)
    verts, nverts = verts[keep], nverts[keep]
    verts[nverts == 3, 3] = 0.0
    return np.ascontiguousarray(verts), np.ascontiguousarray(nverts)


def quad_rays(verts, nverts, n=6000, seed=4):
    rng = np.random.default_rng(seed)
    P = len(nverts)
    p = rng.integers(0, P, n)
    V = verts[p]
    k = nverts[p]
    kind = rng.integers(0, 5, n)
    aim = np.zeros((n, 3))
    for i in range(n):
        v, m = V[i], k[i]
        if kind[i] == 0:
            aim[i] = v[rng.integers(0, m)]                                   # a corner
        elif kind[i] == 1:
            a = rng.integers(0, m); aim[i] = 0.5 * (v[a] + v[(a + 1) % m])   # an edge midpoint
        elif kind[i] == 2:
            aim[i] = 0.5 * (v[0] + v[2])                                     # the diagonal both triangles of a quadrilateral share
        elif kind[i] == 3:
            a = rng.integers(0, m); aim[i] = 0.25 * v[a] + 0.75 * v[(a + 1) % m]
        else:
            w = rng.dirichlet(np.ones(m)); aim[i] = (w[:, None] * v[:m]).sum(0)
    o = rng.uniform(0.05, 0.95, (n, 3)) * np.asarray(SIZE)
    o[::9] = scenes.snap(o[::9])
    d = aim - o
    d[np.linalg.norm(d, axis=1) == 0] = (0.0, 0.0, 1.0)
    d[1::2] /= np.linalg.norm(d[1::2], axis=1, keepdims=True)               # half of them not normalised: t is in units of |d|
    return np.ascontiguousarray(np.concatenate([o, d], 1))


def main():
    verts, nverts = quad_scene()
    rays = quad_rays(verts, nverts)
    T = po.Topology(verts, nverts)
    vox = po.VoxelGrid([T], domain=VOXEL_D, build_mode=0)
    ev_v, _ = vox.shoot(rays)
    e1 = ev_v["poly_id"].astype(np.int32)
    out = {"verts": verts, "nverts": nverts, "rays": rays, "excl1": e1, "voxel": ev_v, "voxel_excl": vox.shoot(rays, excl1=e1)[0],
           "params": np.array([VOXEL_D, OCT[0], OCT[1], KD[0], KD[1]])}
    oc = po.Octree([T], *OCT)
    out["octree"] = oc.shoot(rays)[0]
    out["octree_excl"] = oc.shoot(rays, excl1=e1)[0]
    out["kdtree"] = po.KDTree([T], *KD).shoot(rays)[0]
    path = os.path.join(ROOT, "tests", "golden", "c3_quads.npz")
    np.savez_compressed(path, **out)
    hq = nverts[ev_v["poly_id"][ev_v["hit"] != 0]] == 4
    print(f"wrote {path}: {len(nverts)} polygons ({int((nverts == 4).sum())} quadrilaterals), {len(rays)} rays, "
          f"voxel hits {int(ev_v['hit'].sum())} ({hq.mean():.0%} on quadrilaterals), octree hits {int(out['octree']['hit'].sum())}")


if __name__ == "__main__":
    main()
