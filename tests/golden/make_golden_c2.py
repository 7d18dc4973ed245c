"""Generates tests/golden/c2_ties.npz -- the second committed vector set: exact ties and trees over TWO topologies.

  * Model[0]: the tie scene of tests/test_gpu_ties.py (lattice shoebox + lattice triangles + coplanar overlapping triangles +
    a quarter of the polygons once more, half of those with rotated corners) -- all triangles;
  * Model[1]: a smaller lattice triangle soup in a shifted box (fewer polygons than Model[0], as the reference needs);
  * rays aimed exactly at corners, edge midpoints and edge quarter points of Model[0] (tests/test_gpu_ties.tie_rays).

Expected X_Events from the ORACLE (not from the reference, which cannot be run here: parity unpinned, oracle/hare_oracle.h):
voxel grid over [Model[0]] (plain and with poly_origin1 = the polygon first hit), octree and kd-tree over [Model[0]], and octree
and kd-tree over [Model[0], Model[1]] shot at top_index 0 and 1 ("Octree - alt.cs":63-88,123; KDTree.cs:67-87).

Run from the repo root:  python tests/golden/make_golden_c2.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import hare_amd.scenes as scenes  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from tests.test_gpu_ties import tie_rays, tie_scene  # noqa: E402

VOXEL_D = 8
OCT = (6, 4)
KD = (7, 6)


def second_topology(seed=9, P=300):
    rng = np.random.default_rng(seed)
    c = scenes.snap(rng.uniform(0.5, 3.5, (P, 3)) * [1.5, 1.25, 1.0] + [1.0, 0.5, 0.25])
    a = scenes.snap(rng.uniform(-0.5, 0.5, (P, 3)))
    b = scenes.snap(rng.uniform(-0.5, 0.5, (P, 3)))
    ok = np.linalg.norm(np.cross(a, b), axis=1) > 1e-3
    v = np.zeros((int(ok.sum()), 4, 3))
    v[:, 0] = c[ok]; v[:, 1] = (c + a)[ok]; v[:, 2] = (c + b)[ok]
    return np.ascontiguousarray(v), np.full(len(v), 3, np.int32)


def main():
    v0, n0, size = tie_scene()
    assert np.all(n0 == 3)
    v1, n1 = second_topology()
    assert len(n1) < len(n0)
    rays = tie_rays(v0, n0, size, n=6000)
    T0, T1 = po.Topology(v0, n0), po.Topology(v1, n1)
    vox = po.VoxelGrid([T0], domain=VOXEL_D, build_mode=0)
    ev_v, _ = vox.shoot(rays)
    e1 = ev_v["poly_id"].astype(np.int32)
    ev_vx, _ = vox.shoot(rays, excl1=e1)
    out = {"rays": rays, "excl1": e1, "verts0": v0[:, :3, :], "verts1": v1[:, :3, :], "voxel": ev_v, "voxel_excl": ev_vx,
           "params": np.array([VOXEL_D, OCT[0], OCT[1], KD[0], KD[1]])}
    out["octree"] = po.Octree([T0], *OCT).shoot(rays)[0]
    out["octree_excl"] = po.Octree([T0], *OCT).shoot(rays, excl1=e1)[0]
    out["kdtree"] = po.KDTree([T0], *KD).shoot(rays)[0]
    oc2, kd2 = po.Octree([T0, T1], *OCT), po.KDTree([T0, T1], *KD)
    e1b = np.where(e1 < len(n1), e1, -1).astype(np.int32)
    for top in (0, 1):
        out["octree2_top%d" % top] = oc2.shoot(rays, top_index=top)[0]
        out["octree2_top%d_excl" % top] = oc2.shoot(rays, top_index=top, excl1=e1b)[0]
        out["kdtree2_top%d" % top] = kd2.shoot(rays, top_index=top)[0]
    out["excl1_two"] = e1b
    path = os.path.join(ROOT, "tests", "golden", "c2_ties.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(rays), "rays;", len(n0), "+", len(n1), "triangles;", int(ev_v["hit"].sum()), "voxel hits;",
          os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
