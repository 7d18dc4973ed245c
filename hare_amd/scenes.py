"""Deterministic synthetic scenes and ray sets for the Hare ray-cast path.

Nothing here exists in the reference (it ships no meshes); these are the
harness inputs every backend (oracle, HIP) reads from the SAME arrays.

Rules that make the inputs safe for exact parity:
  * every mesh coordinate is snapped to the 2^-8 m lattice, so Topology's
    Math.Round(x, 15) ingest (Hare_Geometry_Topology.cs:345) leaves it bit-identical and
    no two distinct lattice points share a Hash2 1 mm sub-cell (Hare_Geometry_Primitives.cs:237-250);
  * the shell's min corner is the origin, so the Octree root-box quirk
    ("Octree - alt.cs":78-82, `max + min / 2`) still covers the model;
  * the BASELINE meshes (shoebox, hall, cathedral) are all triangles, polygon index = generation order; `hall_quads` is the hall with
    its flat lattices left UN-SPLIT: planar quadrilaterals (Hare_Geometry_Polygons.cs:731-823) beside the triangles of the two
    displaced surfaces -- what a Pachyderm model looks like (round 5).

Mesh format: verts float64 [P, 4, 3] (corner 3 zero for triangles), nverts int32 [P].
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

LATTICE = 2.0 ** -8


def snap(a):
    return np.round(np.asarray(a, dtype=np.float64) / LATTICE) * LATTICE


@dataclass
class Mesh:
    name: str
    verts: np.ndarray     # [P,4,3] f64
    nverts: np.ndarray    # [P] i32
    size: tuple           # (Lx, Ly, Lz) of the shell

    @property
    def P(self) -> int:
        return int(self.verts.shape[0])


_QUADS = False      # set by hall_quads() while it generates: flat patches come back as quadrilaterals [n,4,3] instead of triangle pairs


def _patch(origin, eu, ev, nu, nv, disp=None):
    """Triangulated rectangle origin + s*eu + t*ev, nu x nv quads, each split along the same
    diagonal.  disp(points[N,3]) -> displaced points, applied before snapping.  (hall_quads: a patch
    without displacement -- every cell of it exactly planar on the lattice -- is left un-split, corners
    p00, p10, p11, p01: the diagonal the triangles use is the one Quadrilateral.Intersect tries first.)"""
    origin = np.asarray(origin, np.float64)
    eu = np.asarray(eu, np.float64)
    ev = np.asarray(ev, np.float64)
    s = np.arange(nu + 1, dtype=np.float64) / nu
    t = np.arange(nv + 1, dtype=np.float64) / nv
    S, T = np.meshgrid(s, t, indexing="ij")
    pts = origin[None, None, :] + S[..., None] * eu[None, None, :] + T[..., None] * ev[None, None, :]
    if disp is not None:
        pts = disp(pts.reshape(-1, 3)).reshape(nu + 1, nv + 1, 3)
    pts = snap(pts)
    p00 = pts[:-1, :-1]
    p10 = pts[1:, :-1]
    p11 = pts[1:, 1:]
    p01 = pts[:-1, 1:]
    if _QUADS and disp is None:
        return np.stack([p00, p10, p11, p01], axis=2).reshape(-1, 4, 3)
    ta = np.stack([p00, p10, p11], axis=2)   # [nu,nv,3,3]
    tb = np.stack([p00, p11, p01], axis=2)
    tris = np.stack([ta, tb], axis=2).reshape(-1, 3, 3)  # face-major, row-major, A then B
    return tris


def _n(length, edge):
    return max(1, int(round(length / edge)))


def _box(lo, hi, edge, skip=()):
    """Six tessellated faces of an axis-aligned box; skip = subset of
    {'x0','x1','y0','y1','z0','z1'}."""
    lo = np.asarray(lo, np.float64)
    hi = np.asarray(hi, np.float64)
    d = hi - lo
    out = []
    nx, ny, nz = _n(d[0], edge), _n(d[1], edge), _n(d[2], edge)
    if "z0" not in skip:
        out.append(_patch(lo, [d[0], 0, 0], [0, d[1], 0], nx, ny))
    if "z1" not in skip:
        out.append(_patch([lo[0], lo[1], hi[2]], [d[0], 0, 0], [0, d[1], 0], nx, ny))
    if "x0" not in skip:
        out.append(_patch(lo, [0, d[1], 0], [0, 0, d[2]], ny, nz))
    if "x1" not in skip:
        out.append(_patch([hi[0], lo[1], lo[2]], [0, d[1], 0], [0, 0, d[2]], ny, nz))
    if "y0" not in skip:
        out.append(_patch(lo, [d[0], 0, 0], [0, 0, d[2]], nx, nz))
    if "y1" not in skip:
        out.append(_patch([lo[0], hi[1], lo[2]], [d[0], 0, 0], [0, 0, d[2]], nx, nz))
    return out


def _finish(name, parts, size):
    vs, ns = [], []
    for part in parts:                       # generation order is kept: polygon index = position
        k = part.shape[1]                    # 3 or 4 corners
        v = np.zeros((part.shape[0], 4, 3), np.float64)
        v[:, :k, :] = part
        vs.append(v)
        ns.append(np.full(part.shape[0], k, np.int32))
    verts, nverts = np.concatenate(vs, axis=0), np.concatenate(ns, axis=0)
    # drop degenerate polygons a snap could create (none expected; guard anyway)
    e1 = verts[:, 1] - verts[:, 0]
    e2 = verts[:, 2] - verts[:, 0]
    keep = np.linalg.norm(np.cross(e1, e2), axis=1) > 0
    return Mesh(name, np.ascontiguousarray(verts[keep]), np.ascontiguousarray(nverts[keep]), size)


def shoebox(nface: int = 9, size=(10.0, 7.0, 4.0)) -> Mesh:
    """S1 'shoebox-1k': 6 faces x nface x nface quads x 2 = 972 triangles at nface = 9."""
    L = np.asarray(size, np.float64)
    parts = []
    parts.append(_patch([0, 0, 0], [L[0], 0, 0], [0, L[1], 0], nface, nface))        # z = 0
    parts.append(_patch([0, 0, L[2]], [L[0], 0, 0], [0, L[1], 0], nface, nface))     # z = Lz
    parts.append(_patch([0, 0, 0], [0, L[1], 0], [0, 0, L[2]], nface, nface))        # x = 0
    parts.append(_patch([L[0], 0, 0], [0, L[1], 0], [0, 0, L[2]], nface, nface))     # x = Lx
    parts.append(_patch([0, 0, 0], [L[0], 0, 0], [0, 0, L[2]], nface, nface))        # y = 0
    parts.append(_patch([0, L[1], 0], [L[0], 0, 0], [0, 0, L[2]], nface, nface))     # y = Ly
    return _finish("shoebox-1k", parts, tuple(size))


def hall(edge: float = 83.0 / 256.0, size=(40.0, 25.0, 18.0)) -> Mesh:
    """S2 'hall-100k': 40 x 25 x 18 m shell with a sinusoidally displaced ceiling, two balcony
    slabs, 12 square columns, a stage box and a raked-floor wedge; 100,908 triangles at the
    default edge (83/256 m)."""
    Lx, Ly, Lz = size
    parts = []
    nx, ny, nz = _n(Lx, edge), _n(Ly, edge), _n(Lz, edge)
    kx, ky = 2 * math.pi * 3 / Lx, 2 * math.pi * 2 / Ly

    def ceil_disp(p):
        q = p.copy()
        q[:, 2] = q[:, 2] + 0.3 * np.sin(kx * q[:, 0]) * np.sin(ky * q[:, 1])
        return q

    parts.append(_patch([0, 0, 0], [Lx, 0, 0], [0, Ly, 0], nx, ny))                   # floor
    parts.append(_patch([0, 0, Lz], [Lx, 0, 0], [0, Ly, 0], nx, ny, ceil_disp))       # ceiling
    parts.append(_patch([0, 0, 0], [0, Ly, 0], [0, 0, Lz], ny, nz))                   # x = 0
    parts.append(_patch([Lx, 0, 0], [0, Ly, 0], [0, 0, Lz], ny, nz))                  # x = Lx
    parts.append(_patch([0, 0, 0], [Lx, 0, 0], [0, 0, Lz], nx, nz))                   # y = 0
    parts.append(_patch([0, Ly, 0], [Lx, 0, 0], [0, 0, Lz], nx, nz))                  # y = Ly
    # balcony slabs along the two long walls
    parts += _box([4.0, 0.0, 6.0], [36.0, 3.0, 6.5], edge, skip=("y0",))
    parts += _box([4.0, 22.0, 6.0], [36.0, 25.0, 6.5], edge, skip=("y1",))
    # 12 square columns under the balcony fronts
    for k in range(6):
        x0 = 5.5 + 5.75 * k
        parts += _box([x0, 2.5, 0.0], [x0 + 0.625, 3.125, 6.0], edge, skip=("z0", "z1"))
        parts += _box([x0, 21.875, 0.0], [x0 + 0.625, 22.5, 6.0], edge, skip=("z0", "z1"))
    # stage box at the x = Lx end
    parts += _box([33.0, 6.0, 0.0], [40.0, 19.0, 1.25], edge, skip=("z0", "x1"))
    # raked floor wedge rising towards x = 0
    ry = _n(13.0, edge)
    rx = _n(14.0, edge)

    def rake(p):
        q = p.copy()
        q[:, 2] = (16.0 - q[:, 0]) * (2.5 / 14.0)
        return q

    parts.append(_patch([2.0, 6.0, 0.0], [14.0, 0, 0], [0, 13.0, 0], rx, ry, rake))
    parts.append(_patch([2.0, 6.0, 0.0], [0, 13.0, 0], [0, 0, 2.5], ry, _n(2.5, edge)))  # riser at x = 2
    return _finish("hall-100k", parts, tuple(size))


def cathedral(edge: float = 51.0 / 256.0, size=(90.0, 40.0, 35.0)) -> Mesh:
    """S3 'cathedral-1M': 90 x 40 x 35 m shell, barrel-vault ceiling, 2 x 14 columns; 986,416
    triangles at the default edge (51/256 m)."""
    Lx, Ly, Lz = size
    parts = []
    nx, ny, nz = _n(Lx, edge), _n(Ly, edge), _n(Lz, edge)
    Rv = 32.0

    def vault(p):
        q = p.copy()
        yy = q[:, 1] - Ly / 2
        q[:, 2] = Lz - (Rv - np.sqrt(Rv * Rv - yy * yy))
        return q

    parts.append(_patch([0, 0, 0], [Lx, 0, 0], [0, Ly, 0], nx, ny))
    parts.append(_patch([0, 0, Lz], [Lx, 0, 0], [0, Ly, 0], nx, ny, vault))
    parts.append(_patch([0, 0, 0], [0, Ly, 0], [0, 0, Lz], ny, nz))
    parts.append(_patch([Lx, 0, 0], [0, Ly, 0], [0, 0, Lz], ny, nz))
    parts.append(_patch([0, 0, 0], [Lx, 0, 0], [0, 0, Lz], nx, nz))
    parts.append(_patch([0, Ly, 0], [Lx, 0, 0], [0, 0, Lz], nx, nz))
    for k in range(14):
        x0 = 6.0 + 6.0 * k
        parts += _box([x0, 9.0, 0.0], [x0 + 1.25, 10.25, 24.0], edge, skip=("z0",))
        parts += _box([x0, 29.75, 0.0], [x0 + 1.25, 31.0, 24.0], edge, skip=("z0",))
    return _finish("cathedral-1M", parts, tuple(size))


def hall_quads(edge: float = 83.0 / 256.0, size=(40.0, 25.0, 18.0)) -> Mesh:
    """The hall with its flat lattices un-split: 39,263 planar quadrilaterals (walls, floor, balconies, columns, stage, riser) + the
    22,382 triangles of the displaced ceiling and the raked floor = 61,645 polygons covering exactly the surfaces of `hall`."""
    global _QUADS
    _QUADS = True
    try:
        m = hall(edge, size)
    finally:
        _QUADS = False
    m.name = "hall-quads-62k"
    return m


SCENES = {"shoebox": shoebox, "hall": hall, "cathedral": cathedral, "hall_quads": hall_quads}


# ---------------------------------------------------------------- rays
def burst_rays(n: int, size, start: int = 0, count: int | None = None) -> np.ndarray:
    """Spherical-Fibonacci burst from (0.31 Lx, 0.42 Ly, 0.37 Lz); rows [x,y,z,dx,dy,dz].
    start/count select a contiguous shard of the global n-ray burst (multi-GPU)."""
    if count is None:
        count = n - start
    i = np.arange(start, start + count, dtype=np.float64)
    z = 1.0 - (2.0 * i + 1.0) / n
    phi = i * (math.pi * (3.0 - math.sqrt(5.0)))
    r = np.sqrt(np.maximum(0.0, 1.0 - z * z))
    rays = np.empty((count, 6), np.float64)
    rays[:, 0] = 0.31 * size[0]
    rays[:, 1] = 0.42 * size[1]
    rays[:, 2] = 0.37 * size[2]
    rays[:, 3] = r * np.cos(phi)
    rays[:, 4] = r * np.sin(phi)
    rays[:, 5] = z
    return rays


def _splitmix64(seed: int, n: int) -> np.ndarray:
    """n outputs of splitmix64 seeded with `seed` (counter form, vectorised)."""
    with np.errstate(over="ignore"):
        k = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + k * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _u01(bits: np.ndarray) -> np.ndarray:
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def random_rays(n: int, size, seed: int = 0x48617265, shrink: float = 0.01) -> np.ndarray:
    """Config-1 rays: origins uniform in the box shrunk by `shrink`, directions uniform on the
    sphere; splitmix64 seeded with 'Hare'."""
    u = _u01(_splitmix64(seed, 5 * n)).reshape(n, 5)
    rays = np.empty((n, 6), np.float64)
    for a in range(3):
        rays[:, a] = shrink + u[:, a] * (size[a] - 2 * shrink)
    z = 2.0 * u[:, 3] - 1.0
    phi = 2.0 * math.pi * u[:, 4]
    r = np.sqrt(np.maximum(0.0, 1.0 - z * z))
    rays[:, 3] = r * np.cos(phi)
    rays[:, 4] = r * np.sin(phi)
    rays[:, 5] = z
    return rays
