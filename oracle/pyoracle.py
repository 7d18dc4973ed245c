"""ctypes binding of oracle/_build/libhare_oracle.so -- the CPU restatement of Hare's ray-cast path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg of
bench.py, as the checker / the timed CPU baseline.  The product package (hare_amd) never imports it.
PARITY UNPINNED (see oracle/hare_oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libhare_oracle.so")

XEVENT_DTYPE = np.dtype(
    [("t", "<f8"), ("u", "<f8"), ("v", "<f8"), ("x", "<f8"), ("y", "<f8"), ("z", "<f8"),
     ("poly_id", "<i4"), ("hit", "<i4")]
)
assert XEVENT_DTYPE.itemsize == 56


class Counters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("hits", C.c_uint64), ("cells", C.c_uint64),
                ("entries", C.c_uint64), ("tests", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class TopologyC(C.Structure):
    _fields_ = [("P", C.c_int32), ("verts", C.c_void_p), ("nverts", C.c_void_p),
                ("normals", C.c_void_p), ("min", C.c_double * 3), ("max", C.c_double * 3)]


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
        L.ho_dotnet_round.restype = dbl
        L.ho_dotnet_round.argtypes = [dbl, C.c_int]
        L.ho_polygon_normals.argtypes = [vp, vp, i32, vp]
        L.ho_finish_topology_bounds.argtypes = [vp, vp, i32, vp, vp]
        L.ho_polygon_centroids.argtypes = [vp, vp, i32, vp]
        L.ho_build_topology.restype = i32
        L.ho_build_topology.argtypes = [vp, vp, i32, vp]
        L.ho_poly_box_overlap.argtypes = [vp, vp, vp, i32]
        L.ho_aabb_intersect_move.argtypes = [vp, vp, vp, vp]
        L.ho_poly_intersect_fast.argtypes = [vp, i32, vp, vp, vp, vp, vp]
        L.ho_poly_intersect_full.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp]
        L.ho_voxel_build.restype = vp
        L.ho_voxel_build.argtypes = [vp, i32, i32, C.c_int]
        L.ho_voxel_build_adaptive.restype = vp
        L.ho_voxel_build_adaptive.argtypes = [vp, i32, i32, i32]
        L.ho_voxel_free.argtypes = [vp]
        L.ho_voxel_ct.restype = i32
        L.ho_voxel_ct.argtypes = [vp]
        L.ho_voxel_char_step.restype = dbl
        L.ho_voxel_char_step.argtypes = [vp]
        L.ho_voxel_geometry.argtypes = [vp, vp, vp, vp]
        L.ho_voxel_cell_start.restype = vp
        L.ho_voxel_cell_start.argtypes = [vp, i32]
        L.ho_voxel_cell_items.restype = vp
        L.ho_voxel_cell_items.argtypes = [vp, i32]
        L.ho_voxel_box.argtypes = [vp, i32, i32, i32, vp, vp]
        L.ho_voxel_shoot_batch.argtypes = [vp, vp, i32, i64, vp, vp, vp, i32, C.c_int, C.c_int, vp, vp]
        L.ho_voxel_pool_new.restype = vp
        L.ho_voxel_pool_new.argtypes = [vp, vp]
        L.ho_voxel_pool_free.argtypes = [vp]
        L.ho_voxel_pool_shoot.argtypes = [vp, vp, i32, i32, i32, i32, vp]
        L.ho_last_error.restype = C.c_char_p
        L.ho_test_fail_alloc_after.argtypes = [C.c_int]
        L.ho_test_stack_cap.argtypes = [C.c_int]
        L.ho_last_error.argtypes = []
        L.ho_octree_build.restype = vp
        L.ho_octree_build.argtypes = [vp, i32, i32, i32]
        L.ho_octree_free.argtypes = [vp]
        L.ho_octree_node_count.restype = i32
        L.ho_octree_node_count.argtypes = [vp]
        L.ho_octree_item_total.restype = i64
        L.ho_octree_item_total.argtypes = [vp]
        L.ho_octree_export.argtypes = [vp, vp, vp, vp, vp, vp]
        L.ho_octree_shoot_batch.argtypes = [vp, vp, i32, i64, vp, vp, vp, C.c_int, vp, vp]
        L.ho_kdtree_build.restype = vp
        L.ho_kdtree_build.argtypes = [vp, i32, i32, i32]
        L.ho_kdtree_free.argtypes = [vp]
        L.ho_kdtree_node_count.restype = i32
        L.ho_kdtree_node_count.argtypes = [vp]
        L.ho_kdtree_item_total.restype = i64
        L.ho_kdtree_item_total.argtypes = [vp]
        L.ho_kdtree_export.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.ho_kdtree_shoot_batch.argtypes = [vp, vp, i32, i64, vp, vp, vp, i32, C.c_int, vp, vp]
        L.ho_brute_shoot.argtypes = [vp, vp, i32, i32, C.c_int, vp]
        L.ho_reflect.argtypes = [vp, vp, vp, vp]
        L.ho_reflect_batch.argtypes = [vp, i64, vp, vp, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data if a is not None else None


def _opt_i32(a, n):
    if a is None:
        return None
    a = np.ascontiguousarray(a, np.int32)
    assert a.shape == (n,)
    return a


class Topology:
    """Flattened Hare.Geometry.Topology as a host would read it back from the managed object."""

    def __init__(self, verts, nverts, normals=None, tmin=None, tmax=None, ingest: bool = False):
        L = lib()
        verts = np.ascontiguousarray(verts, np.float64).reshape(-1, 4, 3)
        self.nverts = np.ascontiguousarray(nverts, np.int32)
        self.P = int(verts.shape[0])
        if ingest:  # Topology(Point[][]): Math.Round + Hash2 dedupe
            out = np.zeros_like(verts)
            self.vertex_count = int(L.ho_build_topology(_p(verts), _p(self.nverts), self.P, _p(out)))
            verts = out
        self.verts = verts
        if normals is None:
            normals = np.zeros((self.P, 3), np.float64)
            L.ho_polygon_normals(_p(self.verts), _p(self.nverts), self.P, _p(normals))
        self.normals = np.ascontiguousarray(normals, np.float64)
        if tmin is None or tmax is None:
            tmin = np.zeros(3)
            tmax = np.zeros(3)
            L.ho_finish_topology_bounds(_p(self.verts), _p(self.nverts), self.P, _p(tmin), _p(tmax))
        self.min = np.ascontiguousarray(tmin, np.float64)
        self.max = np.ascontiguousarray(tmax, np.float64)

    def c_struct(self) -> TopologyC:
        t = TopologyC()
        t.P = self.P
        t.verts = _p(self.verts)
        t.nverts = _p(self.nverts)
        t.normals = _p(self.normals)
        for a in range(3):
            t.min[a] = self.min[a]
            t.max[a] = self.max[a]
        return t


class _Models:
    def __init__(self, topos):
        self.topos = list(topos)
        self.arr = (TopologyC * len(self.topos))(*[t.c_struct() for t in self.topos])
        self.M = len(self.topos)


class VoxelGrid:
    """Oracle Voxel_Grid (Voxel_Grid.cs)."""

    def __init__(self, topos, domain=None, max_domain=None, avg_polys=None, build_mode: int = 1):
        L = lib()
        self.models = _Models(topos)
        if domain is not None:
            self.h = L.ho_voxel_build(self.models.arr, self.models.M, int(domain), int(build_mode))
        else:
            self.h = L.ho_voxel_build_adaptive(self.models.arr, self.models.M, int(max_domain), int(avg_polys))
        _built(self.h)
        self.ct = int(L.ho_voxel_ct(self.h))
        self.char_step = float(L.ho_voxel_char_step(self.h))
        self.obox_min = np.zeros(3)
        self.obox_max = np.zeros(3)
        self.voxel_dims = np.zeros(3)
        L.ho_voxel_geometry(self.h, _p(self.obox_min), _p(self.obox_max), _p(self.voxel_dims))

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.ho_voxel_free(self.h)
            self.h = None

    def lists(self, m: int = 0):
        L = lib()
        n = self.ct ** 3
        start = np.ctypeslib.as_array(C.cast(L.ho_voxel_cell_start(self.h, m), C.POINTER(C.c_uint32)), (n + 1,)).copy()
        tot = int(start[-1])
        items = (np.ctypeslib.as_array(C.cast(L.ho_voxel_cell_items(self.h, m), C.POINTER(C.c_int32)), (max(tot, 1),))[:tot]).copy()
        return start, items

    def box(self, x, y, z):
        mn = np.zeros(3)
        mx = np.zeros(3)
        lib().ho_voxel_box(self.h, x, y, z, _p(mn), _p(mx))
        return mn, mx

    def shoot(self, rays, top_index=0, excl1=None, excl2=None, first_ray_id=1, nthreads=1, mutate=False):
        """Returns (events[n] XEVENT_DTYPE, counters dict[, moved rays if mutate])."""
        L = lib()
        rays = np.array(rays, np.float64, order="C", copy=True).reshape(-1, 6)
        n = rays.shape[0]
        out = np.zeros(n, XEVENT_DTYPE)
        ctr = Counters()
        e1 = _opt_i32(excl1, n)
        e2 = _opt_i32(excl2, n)
        L.ho_voxel_shoot_batch(self.h, self.models.arr, top_index, n, _p(rays), _p(e1), _p(e2),
                               first_ray_id, 0 if mutate else 1, nthreads, _p(out), C.addressof(ctr))
        if mutate:
            return out, ctr.as_dict(), rays
        return out, ctr.as_dict()

    def pool(self):
        return VoxelPool(self)


def _built(handle):
    """A builder returns NULL when an allocation failed or a tree outgrew its budget (hare_oracle.h): an exception, not a
    segmentation fault in the next call."""
    if not handle:
        raise MemoryError((lib().ho_last_error() or b"oracle build failed").decode())


class VoxelPool:
    """Faithful 500-slot mailbox pool (Voxel_Grid.cs:54-62, :334-342), single-threaded."""

    def __init__(self, grid: VoxelGrid):
        self.grid = grid
        self.h = lib().ho_voxel_pool_new(grid.h, grid.models.arr)

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.ho_voxel_pool_free(self.h)
            self.h = None

    def shoot(self, ray, ray_id, top_index=0, po1=-1, po2=-1):
        r = np.array(ray, np.float64, copy=True).reshape(6)
        out = np.zeros(1, XEVENT_DTYPE)
        lib().ho_voxel_pool_shoot(self.h, _p(r), int(ray_id), top_index, po1, po2, _p(out))
        return out[0], r


class Octree:
    """Oracle Octree ("Octree - alt.cs")."""

    def __init__(self, topos, max_depth, max_polys):
        L = lib()
        self.models = _Models(topos)
        self.h = L.ho_octree_build(self.models.arr, self.models.M, int(max_depth), int(max_polys))
        _built(self.h)
        self.n_nodes = int(L.ho_octree_node_count(self.h))

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.ho_octree_free(self.h)
            self.h = None

    def export(self):
        L = lib()
        n = self.n_nodes
        tot = int(L.ho_octree_item_total(self.h))
        boxes = np.zeros((n, 6))
        fc = np.zeros(n, np.int32)
        st = np.zeros(n, np.int32)
        cn = np.zeros(n, np.int32)
        items = np.zeros(max(tot, 1), np.int32)
        L.ho_octree_export(self.h, _p(boxes), _p(fc), _p(st), _p(cn), _p(items))
        return boxes, fc, st, cn, items[:tot]

    def shoot(self, rays, top_index=0, excl1=None, excl2=None, nthreads=1):
        L = lib()
        rays = np.ascontiguousarray(rays, np.float64).reshape(-1, 6)
        n = rays.shape[0]
        out = np.zeros(n, XEVENT_DTYPE)
        ctr = Counters()
        e1 = _opt_i32(excl1, n)
        e2 = _opt_i32(excl2, n)
        rc = L.ho_octree_shoot_batch(self.h, self.models.arr, top_index, n, _p(rays), _p(e1), _p(e2), nthreads,
                                     _p(out), C.addressof(ctr))
        if rc != 0:       # the checker must never pass a failed trace off as a miss record
            raise MemoryError((L.ho_last_error() or b"oracle octree shoot failed").decode())
        return out, ctr.as_dict()


class KDTree:
    """Oracle KDTree (KDTree.cs)."""

    def __init__(self, topos, max_depth, max_polys):
        L = lib()
        self.models = _Models(topos)
        self.h = L.ho_kdtree_build(self.models.arr, self.models.M, int(max_depth), int(max_polys))
        _built(self.h)
        self.n_nodes = int(L.ho_kdtree_node_count(self.h))

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.ho_kdtree_free(self.h)
            self.h = None

    def export(self):
        L = lib()
        n = self.n_nodes
        tot = int(L.ho_kdtree_item_total(self.h))
        boxes = np.zeros((n, 6))
        split = np.zeros(n)
        axis = np.zeros(n, np.int32)
        left = np.zeros(n, np.int32)
        right = np.zeros(n, np.int32)
        st = np.zeros(n, np.int32)
        cn = np.zeros(n, np.int32)
        items = np.zeros(max(tot, 1), np.int32)
        L.ho_kdtree_export(self.h, _p(boxes), _p(split), _p(axis), _p(left), _p(right), _p(st), _p(cn), _p(items))
        return boxes, split, axis, left, right, st, cn, items[:tot]

    def shoot(self, rays, top_index=0, excl1=None, excl2=None, first_ray_id=1, nthreads=1):
        L = lib()
        rays = np.ascontiguousarray(rays, np.float64).reshape(-1, 6)
        n = rays.shape[0]
        out = np.zeros(n, XEVENT_DTYPE)
        ctr = Counters()
        e1 = _opt_i32(excl1, n)
        e2 = _opt_i32(excl2, n)
        rc = L.ho_kdtree_shoot_batch(self.h, self.models.arr, top_index, n, _p(rays), _p(e1), _p(e2), first_ray_id,
                                     nthreads, _p(out), C.addressof(ctr))
        if rc != 0:
            raise MemoryError((L.ho_last_error() or b"oracle kd-tree shoot failed").decode())
        return out, ctr.as_dict()


def brute(topo: Topology, rays, excl1=None, excl2=None, full_uv=False):
    L = lib()
    rays = np.ascontiguousarray(rays, np.float64).reshape(-1, 6)
    n = rays.shape[0]
    out = np.zeros(n, XEVENT_DTYPE)
    t = topo.c_struct()
    for i in range(n):
        L.ho_brute_shoot(C.addressof(t), rays[i].ctypes.data, -1 if excl1 is None else int(excl1[i]),
                         -1 if excl2 is None else int(excl2[i]), 1 if full_uv else 0, out[i:i + 1].ctypes.data)
    return out


def reflect(topo: Topology, rays, events):
    L = lib()
    rays = np.ascontiguousarray(rays, np.float64).reshape(-1, 6)
    out = np.zeros_like(rays)
    t = topo.c_struct()
    for i in range(rays.shape[0]):
        if events[i]["hit"]:
            L.ho_reflect(C.addressof(t), rays[i].ctypes.data, events[i:i + 1].ctypes.data, out[i].ctypes.data)
    return out


def reflect_batch(topo: Topology, rays, events):
    """Specular bounce of a whole batch; rays that missed keep their record."""
    rays = np.ascontiguousarray(rays, np.float64).reshape(-1, 6)
    events = np.ascontiguousarray(events)
    assert events.dtype == XEVENT_DTYPE and len(events) == len(rays)
    out = np.empty_like(rays)
    t = topo.c_struct()
    lib().ho_reflect_batch(C.addressof(t), rays.shape[0], _p(rays), _p(events), _p(out))
    return out


def poly_box_overlap(bmin, bmax, poly_verts) -> bool:
    bmin = np.ascontiguousarray(bmin, np.float64)
    bmax = np.ascontiguousarray(bmax, np.float64)
    pv = np.ascontiguousarray(poly_verts, np.float64).reshape(-1, 3)
    return bool(lib().ho_poly_box_overlap(_p(bmin), _p(bmax), _p(pv), pv.shape[0]))


def aabb_intersect_move(bmin, bmax, ray):
    bmin = np.ascontiguousarray(bmin, np.float64)
    bmax = np.ascontiguousarray(bmax, np.float64)
    r = np.array(ray, np.float64, copy=True).reshape(6)
    t = C.c_double(0)
    ok = lib().ho_aabb_intersect_move(_p(bmin), _p(bmax), _p(r), C.addressof(t))
    return bool(ok), t.value, r


def dotnet_round(x: float, digits: int = 15) -> float:
    return float(lib().ho_dotnet_round(float(x), int(digits)))
