"""Known-answer tests that pin the oracle (the CPU restatement of Hare's ray-cast path).

The reference ships no tests or vectors and cannot be run here, so the pins are
closed forms: a shoebox 8 x 4 x 2 m with 8 x 8 quads per face has quad edges 1, 0.5 and 0.25 m;
every coordinate, edge, cross product and determinant in the Moller-Trumbore test is then a small
dyadic rational and every determinant a power of two, so binary64 evaluates RayXtri EXACTLY and
`t`, `X_Point` must equal the geometric answer bit for bit, whatever the operation order.
"""
import numpy as np
import pytest

import hare_amd.scenes as scenes
from oracle import pyoracle as po

L = (8.0, 4.0, 2.0)
NF = 8


@pytest.fixture(scope="module")
def box():
    m = scenes.shoebox(nface=NF, size=L)
    T = po.Topology(m.verts, m.nverts)
    return m, T


@pytest.fixture(scope="module")
def grid(box):
    return po.VoxelGrid([box[1]], domain=8, build_mode=0)


def face_poly(face, i, j, upper):
    """Polygon index by generation order: face-major, quad (i,j) row-major, A (s>=t) then B."""
    return face * 2 * NF * NF + (i * NF + j) * 2 + (1 if upper else 0)


# faces in generator order: 0 z=0, 1 z=Lz, 2 x=0, 3 x=Lx, 4 y=0, 5 y=Ly
# (face, axis of travel, sign, (u axis, v axis)) -- patch parameter u runs along the first edge
CASES = [
    (3, 0, +1, (1, 2)), (2, 0, -1, (1, 2)),
    (5, 1, +1, (0, 2)), (4, 1, -1, (0, 2)),
    (1, 2, +1, (0, 1)), (0, 2, -1, (0, 1)),
]


@pytest.mark.parametrize("face,axis,sign,uv", CASES)
def test_axis_aligned_walls_exact(box, grid, face, axis, sign, uv):
    m, T = box
    o = np.array([3.0, 1.0, 0.75])
    # aim at a point strictly inside triangle A (s > t) and one inside B (t > s) of some quad
    for upper, (ds, dt) in ((False, (0.75, 0.25)), (True, (0.25, 0.75))):
        quad = np.array([L[uv[0]] / NF, L[uv[1]] / NF])
        i, j = 2, 3
        target_u = (i + ds) * quad[0]
        target_v = (j + dt) * quad[1]
        oo = o.copy()
        oo[uv[0]] = target_u
        oo[uv[1]] = target_v
        d = np.zeros(3)
        d[axis] = sign
        wall = L[axis] if sign > 0 else 0.0
        ev, _ = grid.shoot(np.concatenate([oo, d])[None, :])
        e = ev[0]
        assert e["hit"] == 1
        assert e["poly_id"] == face_poly(face, i, j, upper)
        assert e["t"] == abs(wall - oo[axis])          # exact
        hp = oo.copy()
        hp[axis] = wall
        assert (e["x"], e["y"], e["z"]) == tuple(hp)    # exact
        assert e["u"] == 0.0 and e["v"] == 0.0          # Voxel_Grid returns u = v = 0 (Voxel_Grid.cs:696-697)


def test_direction_length_scales_t(box, grid):
    # t is in units of |d| (Primitives.cs:470): doubling d halves t exactly
    r = np.array([[3.0, 1.0, 0.75, 2.0, 0.0, 0.0]])
    ev, _ = grid.shoot(r)
    assert ev[0]["t"] == 2.5 and ev[0]["x"] == 8.0


def test_diagonal_ray_exact(box, grid):
    # (1,1,0.5) along (1,1,0): reaches y = 4 at t = 3 -> point (4,4,0.5) on face 5 (y = Ly)
    r = np.array([[1.0, 1.0, 0.5, 1.0, 1.0, 0.0]])
    ev, _ = grid.shoot(r)
    e = ev[0]
    assert e["hit"] == 1 and e["t"] == 3.0
    assert (e["x"], e["y"], e["z"]) == (4.0, 4.0, 0.5)
    # face 5 patch: u along x (quad 1.0), v along z (quad 0.25): x=4.0 is the line between quads
    # i=3|4, z=0.5 between j=1|2 -> the hit is a shared vertex; winner is the first polygon tested
    # with that t: ascending index within the first cell that lists any of them
    b = po.brute(box[1], r)[0]
    assert b["t"] == 3.0
    cands = [p for p in range(box[1].P) if _hits_at(box[1], p, r[0], 3.0)]
    assert e["poly_id"] in cands and b["poly_id"] == min(cands)


def _hits_at(T, p, ray, t):
    one = po.Topology(T.verts[p:p + 1], T.nverts[p:p + 1])
    ev = po.brute(one, ray[None, :])[0]
    return ev["hit"] == 1 and ev["t"] == t


def test_tie_on_quad_diagonal_goes_to_lower_index(box, grid):
    # hit exactly on the A|B diagonal of quad (2,3) of face 3 (s == t): both triangles return the
    # same exact t; the accept is the strict `t < tmin` (Voxel_Grid.cs:693) so A (lower index) stays
    quad = np.array([L[1] / NF, L[2] / NF])
    y = (2 + 0.5) * quad[0]
    z = (3 + 0.5) * quad[1]
    r = np.array([[3.0, y, z, 1.0, 0.0, 0.0]])
    ev, _ = grid.shoot(r)
    assert ev[0]["hit"] == 1 and ev[0]["t"] == 5.0
    assert ev[0]["poly_id"] == face_poly(3, 2, 3, False)


def test_ray_parallel_to_faces_dx_zero(box, grid):
    # dx = dy = +0: tMaxX = tMaxY = +inf, only Z steps (Voxel_Grid.cs:589-632)
    r = np.array([[3.25, 1.125, 0.75, 0.0, 0.0, 1.0]])
    ev, ctr = grid.shoot(r)
    assert ev[0]["hit"] == 1 and ev[0]["t"] == 1.25 and ev[0]["z"] == 2.0


def test_origin_outside_grid_moves_ray_and_adds_t_start(box, grid):
    # F11: AABB.Intersect(ref R, ref tmin) moves R to the OBox entry; t returned = tmin + t_start
    r = np.array([[-5.0, 1.125, 0.75, 1.0, 0.0, 0.0]])
    ev, _, moved = grid.shoot(r, mutate=True)
    e = ev[0]
    assert e["hit"] == 1
    assert e["poly_id"] // (2 * NF * NF) == 2           # the x = 0 wall
    assert abs(e["t"] - 5.0) < 1e-12 and e["x"] == 0.0  # hit point from the moved origin
    t_start = (grid.obox_min[0] - (-5.0)) * (1 / 1.0)   # AABB_Main.cs:187-188
    assert moved[0, 0] == -5.0 + 1.0 * t_start          # AABB_Main.cs:255: origin now on OBox.Min.x (to an ulp)
    assert abs(moved[0, 0] - grid.obox_min[0]) < 1e-15
    assert moved[0, 1] == 1.125 and moved[0, 2] == 0.75
    # from outside and pointing away: miss, X_Event() record
    ev2, _ = grid.shoot(np.array([[-5.0, 1.0, 1.0, -1.0, 0.0, 0.0]]))
    assert ev2[0]["hit"] == 0 and ev2[0]["poly_id"] == -1 and ev2[0]["t"] == 0.0


def test_exclusion_overload_skips_origin_polygons(box, grid):
    # Shoot(R, top, out ev, poly_origin1, poly_origin2): Voxel_Grid.cs:351,477
    r = np.array([[3.0, 1.3125, 0.8125, 1.0, 0.0, 0.0]])
    ev, _ = grid.shoot(r)
    first = int(ev[0]["poly_id"])
    ev2, _ = grid.shoot(r, excl1=[first])
    assert ev2[0]["hit"] == 0   # nothing else lies on this ray (the far wall is the only crossing)
    # excluding via the second slot behaves the same
    ev3, _ = grid.shoot(r, excl1=[-1], excl2=[first])
    assert ev3[0]["hit"] == 0


def test_mailbox_ray_id_zero_always_misses(box, grid):
    # F7(a): fresh mailbox arrays are zero, the test is `!= R.Ray_ID`, so Ray_ID == 0 skips every polygon
    pool = grid.pool()
    ev, _ = pool.shoot([3.0, 1.0, 0.75, 1.0, 0.0, 0.0], ray_id=0)
    assert ev["hit"] == 0 and ev["poly_id"] == -1
    ev, _ = pool.shoot([3.0, 1.0, 0.75, 1.0, 0.0, 0.0], ray_id=7)
    assert ev["hit"] == 1 and ev["t"] == 5.0


def test_mailbox_stale_slot_collision(box, grid):
    # F7: a later ray that reuses a Ray_ID in the SAME slot (500 shoots later) sees stale marks
    pool = grid.pool()
    ray = [3.0, 1.0, 0.75, 1.0, 0.0, 0.0]
    ev, _ = pool.shoot(ray, ray_id=42)          # slot 1
    assert ev["hit"] == 1
    for _ in range(499):                         # slots 2..499, 0
        pool.shoot(ray, ray_id=9999)
    ev, _ = pool.shoot(ray, ray_id=42)          # slot 1 again, same id -> every polygon already "tested"
    assert ev["hit"] == 0


def test_negative_zero_direction_marches_out_with_pending_hit():
    # F12 + the -0.0 branch of Voxel_Grid.cs:604-617: dy = -0.0 takes the `else` branch, tMaxY = -inf,
    # the DDA marches +y out of the grid and Shoot returns a MISS although a hit is pending.
    m = scenes.shoebox(nface=1, size=L)          # two big triangles per wall
    T = po.Topology(m.verts, m.nverts)
    g = po.VoxelGrid([T], domain=2, build_mode=0)
    base = [5.0, 1.0, 0.5, 1.0, 0.0, 0.25]
    ev_pos, _ = g.shoot(np.array([base]))
    assert ev_pos[0]["hit"] == 1 and ev_pos[0]["t"] == 3.0 and ev_pos[0]["x"] == 8.0 and ev_pos[0]["z"] == 1.25
    neg = list(base)
    neg[4] = -0.0
    ev_neg, _ = g.shoot(np.array([neg]))
    assert ev_neg[0]["hit"] == 0 and ev_neg[0]["poly_id"] == -1


def test_voxel_box_arithmetic(box, grid):
    # A.2: Min.c = (i*VoxelDims.c - 0.001) + OBox.Min.c, Max.c = ((i+1)*VoxelDims.c + 0.001) + OBox.Min.c
    vd, om = grid.voxel_dims, grid.obox_min
    mn, mx = grid.box(3, 2, 1)
    for a, i in enumerate((3, 2, 1)):
        assert mn[a] == (i * vd[a] - 0.001) + om[a]
        assert mx[a] == ((i + 1) * vd[a] + 0.001) + om[a]
    # OBox: model bounds -/+ 1e-12 (Finish_Topology), -/+ 0.001 (Epsilon), -/+ 0.1
    assert om[0] == ((0.0 - 0.000000000001) - 0.001) - .1
    assert grid.obox_max[0] == ((8.0 + 0.000000000001) + 0.001) + .1
    assert grid.char_step == min(vd)


def test_normals_follow_polygon_ctor(box):
    # Cross(V1-V0, V2-V0) normalised by three divisions; y component is written -(ax*bz - az*bx),
    # so an exactly-zero y comes out as -0.0 (Hare_Geometry_Math.cs:62-65)
    _, T = box
    n = T.normals[0]   # floor triangle A: (p00, p10, p11) -> +z
    assert n[2] == 1.0 and n[0] == 0.0 and n[1] == 0.0 and np.signbit(n[1])


def test_dotnet_round_and_lattice():
    # F10: Math.Round(x, 15) = rint(x*1e15)/1e15 changes arbitrary doubles but never lattice values
    assert po.dotnet_round(0.1 + 0.2, 15) == 0.3
    rng = np.random.default_rng(1)
    vals = np.round(rng.uniform(-1024, 1024, 2000) * 256) / 256
    assert all(po.dotnet_round(v, 15) == v for v in vals)
    changed = sum(po.dotnet_round(v, 15) != v for v in rng.uniform(0, 40, 2000))
    assert changed > 0


def test_topology_ingest_dedupes_within_1mm_subcell():
    # Hash2 (Primitives.cs:237-250): same 1 m bucket + same 1 mm sub-cell -> the first vertex wins
    v = np.zeros((2, 4, 3))
    v[0, :3] = [[0, 0, 0], [1, 0, 0], [0, 1, 0]]
    v[1, :3] = [[1.0002, 0.0003, 0.0], [2, 0, 0], [1, 1, 0]]   # first corner within 1 mm of (1,0,0)
    T = po.Topology(v, np.array([3, 3], np.int32), ingest=True)
    assert tuple(T.verts[1, 0]) == (1.0, 0.0, 0.0)
    assert T.vertex_count == 5


def test_poly_box_overlap_known_cases():
    tri = [[0, 0, 0], [1, 0, 0], [0, 1, 0]]
    assert po.poly_box_overlap([-0.5, -0.5, -0.5], [0.25, 0.25, 0.5], tri)
    assert not po.poly_box_overlap([2, 2, -0.5], [3, 3, 0.5], tri)            # beyond the AABB
    assert not po.poly_box_overlap([0.75, 0.75, -0.5], [1.0, 1.0, 0.5], tri)  # inside the AABB, past the hypotenuse
    assert not po.poly_box_overlap([0, 0, 0.25], [1, 1, 0.5], tri)            # plane test
    assert po.poly_box_overlap([0.5, 0.5, 0.0], [1, 1, 1], tri)               # touches at the hypotenuse midpoint
    quad = [[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]]
    assert po.poly_box_overlap([0.75, 0.75, -0.5], [1.0, 1.0, 0.5], quad)     # second fan triangle


def test_aabb_intersect_move_known():
    ok, t, r = po.aabb_intersect_move([0, 0, 0], [2, 2, 2], [-2, 1, 1, 2, 0, 0])
    assert ok and t == 1.0 and tuple(r[:3]) == (0.0, 1.0, 1.0)
    ok, t, r = po.aabb_intersect_move([0, 0, 0], [2, 2, 2], [-2, 3, 1, 1, 0, 0])   # parallel, outside slab
    assert not ok
    ok, t, r = po.aabb_intersect_move([0, 0, 0], [2, 2, 2], [1, 1, 1, 1, 0, 0])    # origin inside: tmin stays 0
    assert ok and t == 0.0 and tuple(r[:3]) == (1.0, 1.0, 1.0)
