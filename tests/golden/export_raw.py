"""Writes the committed golden vectors (tests/golden/c1_shoebox.npz) as raw little-endian files a .NET program can
read with BinaryReader -- the input of bindings/csharp/tests/GoldenParity.cs, which runs the REFERENCE Hare classes
(Voxel_Grid / Octree / KDTree .Shoot) on the same mesh and rays and diffs their X_Events against these records bit
for bit.  That run is the one step that turns "parity unpinned" into a pinned oracle; it needs a .NET SDK, which the
build container does not have, so it is prepared here and run by whoever has `dotnet`.

    python tests/golden/export_raw.py [out_dir]          # default: tests/golden/raw (git-ignored)

Files (all little-endian, no headers):
    params.txt          D OD OP KDD KDP P N   (grid domain; octree depth, max polys; kd depth, max polys; triangles; rays)
    tris.f64            P x 3 x 3   triangle corners (config 1 shoebox, hare_amd.scenes.shoebox(); polygon index = order)
    rays.f64            N x 6       x y z dx dy dz   (first 10 000: the seeded random rays, then the edge-case rays)
    excl1.i32           N           poly_origin1 of the exclusion overload (= Poly_id of the first voxel hit, -1 on a miss)
    <name>.xev          N x 56 B    X_Event records {t,u,v,x,y,z: f64; Poly_id, Hit: i32} for name in
                        voxel, voxel_excl, octree, octree_excl, kdtree  (a miss is all zeros with Poly_id = -1)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import hare_amd.scenes as scenes  # noqa: E402


def main(out_dir=None):
    out_dir = out_dir or os.path.join(ROOT, "tests", "golden", "raw")
    os.makedirs(out_dir, exist_ok=True)
    G = np.load(os.path.join(ROOT, "tests", "golden", "c1_shoebox.npz"))
    mesh = scenes.shoebox()
    tris = np.ascontiguousarray(mesh.verts[:, :3, :], "<f8")
    rays = np.ascontiguousarray(G["rays"], "<f8")
    n = rays.shape[0]
    tris.tofile(os.path.join(out_dir, "tris.f64"))
    rays.tofile(os.path.join(out_dir, "rays.f64"))
    np.ascontiguousarray(G["excl1"], "<i4").tofile(os.path.join(out_dir, "excl1.i32"))
    for name in ("voxel", "voxel_excl", "octree", "octree_excl", "kdtree"):
        ev = np.ascontiguousarray(G[name])
        assert ev.dtype.itemsize == 56 and len(ev) == n
        ev.tofile(os.path.join(out_dir, name + ".xev"))
    with open(os.path.join(out_dir, "params.txt"), "w") as f:
        f.write(" ".join(str(int(x)) for x in G["params"]) + f" {tris.shape[0]} {n}\n")
    print(f"wrote {out_dir}: {tris.shape[0]} triangles, {n} rays, 5 event files")
    export_c2(os.path.join(out_dir, "c2"))
    export_c3(os.path.join(out_dir, "c3"))
    return out_dir


def export_c3(out_dir):
    """Case 3 (tests/golden/c3_quads.npz): quadrilaterals.  polys.f64 is P x 4 x 3 (corner 3 unused for a triangle), nverts.i32 the corner
    counts (3 or 4); rays.f64, excl1.i32 and the events voxel / voxel_excl / octree / octree_excl / kdtree as in case 1.
    params.txt: D OD OP KDD KDP P N."""
    os.makedirs(out_dir, exist_ok=True)
    G = np.load(os.path.join(ROOT, "tests", "golden", "c3_quads.npz"))
    n = G["rays"].shape[0]
    np.ascontiguousarray(G["verts"], "<f8").tofile(os.path.join(out_dir, "polys.f64"))
    np.ascontiguousarray(G["nverts"], "<i4").tofile(os.path.join(out_dir, "nverts.i32"))
    np.ascontiguousarray(G["rays"], "<f8").tofile(os.path.join(out_dir, "rays.f64"))
    np.ascontiguousarray(G["excl1"], "<i4").tofile(os.path.join(out_dir, "excl1.i32"))
    for name in ("voxel", "voxel_excl", "octree", "octree_excl", "kdtree"):
        ev = np.ascontiguousarray(G[name])
        assert ev.dtype.itemsize == 56 and len(ev) == n
        ev.tofile(os.path.join(out_dir, name + ".xev"))
    with open(os.path.join(out_dir, "params.txt"), "w") as f:
        f.write(" ".join(str(int(x)) for x in G["params"]) + f" {len(G['nverts'])} {n}\n")
    print(f"wrote {out_dir}: {len(G['nverts'])} polygons ({int((G['nverts'] == 4).sum())} quadrilaterals), {n} rays, 5 event files")


def export_c2(out_dir):
    """Case 2 (tests/golden/c2_ties.npz): exact ties + trees over two topologies.  Files as above, plus
    tris0.f64 / tris1.f64 (Model[0], Model[1]), excl1_two.i32, and the events octree2_top{0,1}[_excl].xev, kdtree2_top{0,1}.xev
    of Octree / KDTree built over BOTH topologies and shot at top_index 0 / 1.  params.txt: D OD OP KDD KDP P0 P1 N."""
    os.makedirs(out_dir, exist_ok=True)
    G = np.load(os.path.join(ROOT, "tests", "golden", "c2_ties.npz"))
    n = G["rays"].shape[0]
    np.ascontiguousarray(G["verts0"], "<f8").tofile(os.path.join(out_dir, "tris0.f64"))
    np.ascontiguousarray(G["verts1"], "<f8").tofile(os.path.join(out_dir, "tris1.f64"))
    np.ascontiguousarray(G["rays"], "<f8").tofile(os.path.join(out_dir, "rays.f64"))
    np.ascontiguousarray(G["excl1"], "<i4").tofile(os.path.join(out_dir, "excl1.i32"))
    np.ascontiguousarray(G["excl1_two"], "<i4").tofile(os.path.join(out_dir, "excl1_two.i32"))
    names = ["voxel", "voxel_excl", "octree", "octree_excl", "kdtree"] + [f"{k}2_top{t}{x}" for t in (0, 1) for k, x in
                                                                         (("octree", ""), ("octree", "_excl"), ("kdtree", ""))]
    for name in names:
        ev = np.ascontiguousarray(G[name])
        assert ev.dtype.itemsize == 56 and len(ev) == n
        ev.tofile(os.path.join(out_dir, name + ".xev"))
    with open(os.path.join(out_dir, "params.txt"), "w") as f:
        f.write(" ".join(str(int(x)) for x in G["params"]) + f" {G['verts0'].shape[0]} {G['verts1'].shape[0]} {n}\n")
    print(f"wrote {out_dir}: {G['verts0'].shape[0]} + {G['verts1'].shape[0]} triangles, {n} rays, {len(names)} event files")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
