"""Host-side mirror of the reference's interface for the ray-cast path, over the C-ABI.

Same names, argument meaning and error behaviour as Hare.Geometry's
  Ray / X_Event            Hare_Geometry_Primitives.cs:393-481
  Topology (the members a partition reads)   Hare_Geometry_Topology.cs:418-424, :482, :539, :50/:58
  Spatial_Partition        Spatial_Partition.cs:27-35
  Voxel_Grid / Octree / KDTree constructors  Voxel_Grid.cs:48,128  "Octree - alt.cs":45  KDTree.cs:51
so the parity tests read like calls into the reference.  Every Shoot runs the HIP kernels through
libhare_hip.so; there is no Python or CPU implementation of the path in this package.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import capi
from .capi import KIND_KDTREE, KIND_OCTREE, KIND_VOXEL, RAY_DTYPE, XEVENT_DTYPE, check, lib, ptr


class Ray:
    """Hare.Geometry.Ray (Hare_Geometry_Primitives.cs:393-429)."""

    __slots__ = ("x", "y", "z", "dx", "dy", "dz", "ThreadID", "Ray_ID", "poly_origin1", "poly_origin2")

    def __init__(self, x, y, z, dx, dy, dz, ThreadID_IN: int = 0, ID: int = 0):
        self.x, self.y, self.z = float(x), float(y), float(z)
        self.dx, self.dy, self.dz = float(dx), float(dy), float(dz)
        self.ThreadID, self.Ray_ID = int(ThreadID_IN), int(ID)
        self.poly_origin1 = self.poly_origin2 = 0

    def Reverse(self):
        self.dx *= -1
        self.dy *= -1
        self.dz *= -1


class X_Event:
    """Hare.Geometry.X_Event (Hare_Geometry_Primitives.cs:435-481)."""

    __slots__ = ("u", "v", "t", "Hit", "X_Point", "Poly_id")

    def __init__(self, P=None, u_in=0.0, v_in=0.0, t_in=0.0, Poly_index=-1):
        if P is None:           # X_Event(): :454-462
            self.u = self.v = self.t = 0.0
            self.Hit = False
            self.X_Point = None
            self.Poly_id = -1
        else:                   # X_Event(Point, u, v, t, Poly_index): :472-480
            self.u, self.v, self.t = float(u_in), float(v_in), float(t_in)
            self.Hit = True
            self.X_Point = tuple(float(c) for c in P)
            self.Poly_id = int(Poly_index)

    @staticmethod
    def from_record(r) -> "X_Event":
        if r["hit"]:
            return X_Event((r["x"], r["y"], r["z"]), r["u"], r["v"], r["t"], r["poly_id"])
        return X_Event()


class Topology:
    """The part of Hare.Geometry.Topology a Spatial_Partition reads: polygons with their corner
    coordinates, unit normals and the Min/Max box (after Finish_Topology).  Normals and bounds are
    computed by the library's restatements of the Polygon ctor / Finish_Topology."""

    def __init__(self, verts, nverts=None, normals=None, Min=None, Max=None):
        v = np.ascontiguousarray(verts, np.float64)
        if v.ndim == 3 and v.shape[1] == 3:      # [P,3,3] triangles
            w = np.zeros((v.shape[0], 4, 3), np.float64)
            w[:, :3] = v
            v = w
        self.verts = np.ascontiguousarray(v.reshape(-1, 4, 3))
        P = self.verts.shape[0]
        self.nverts = np.full(P, 3, np.int32) if nverts is None else np.ascontiguousarray(nverts, np.int32)
        if self.nverts.shape != (P,):
            raise ValueError("nverts must have one entry per polygon")
        if normals is None:
            normals = np.zeros((P, 3), np.float64)
            check(lib.hare_polygon_normals(ptr(self.verts), ptr(self.nverts), P, ptr(normals)))
        self.normals = np.ascontiguousarray(normals, np.float64)
        if Min is None or Max is None:
            Min = np.zeros(3)
            Max = np.zeros(3)
            check(lib.hare_topology_bounds(ptr(self.verts), ptr(self.nverts), P, ptr(Min), ptr(Max)))
        self.Min = np.ascontiguousarray(Min, np.float64)
        self.Max = np.ascontiguousarray(Max, np.float64)

    @classmethod
    def from_polygons(cls, T, nverts=None):
        """Topology(Point[][] T) (Hare_Geometry_Topology.cs:120-142) followed by Finish_Topology(): the raw
        corners go through the reference's ingest (Math.Round(x, 15), corners in the same 1 mm Hash2 cell
        merged onto the first one) before normals and bounds are taken.  `T` is [P,3,3], [P,4,3] or a
        sequence of 3- or 4-corner polygons; `nverts` overrides the corner count per polygon.
        The result also carries Vertices_List (`.vertices`) and the per-corner vertex index (`.corner_vertex`)."""
        if not isinstance(T, np.ndarray):
            polys = [np.asarray(p, np.float64).reshape(-1, 3) for p in T]
            soup = np.zeros((len(polys), 4, 3), np.float64)
            nverts = np.zeros(len(polys), np.int32)
            for i, p in enumerate(polys):
                if p.shape[0] not in (3, 4):
                    raise NotImplementedError("Hare Does not yet support polygons of more than 4 sides.")
                soup[i, :p.shape[0]] = p
                nverts[i] = p.shape[0]
        else:
            v = np.ascontiguousarray(T, np.float64)
            soup = np.zeros((v.shape[0], 4, 3), np.float64)
            soup[:, :v.shape[1]] = v
            if nverts is None:
                nverts = np.full(v.shape[0], v.shape[1], np.int32)
        nverts = np.ascontiguousarray(nverts, np.int32)
        P = soup.shape[0]
        verts = np.zeros((P, 4, 3), np.float64)
        corner_vertex = np.full((P, 4), -1, np.int32)
        vertices = np.zeros((max(int(nverts.sum()), 1), 3), np.float64)
        nv = C.c_int32(0)
        rc = lib.hare_topology_ingest(ptr(soup), ptr(nverts), P, ptr(verts), ptr(corner_vertex), ptr(vertices), C.addressof(nv))
        if rc == capi.HARE_E_UNSUPPORTED:
            raise NotImplementedError(capi.last_error())
        check(rc)
        top = cls(verts, nverts)
        top.vertices = vertices[:nv.value].copy()
        top.corner_vertex = corner_vertex
        return top

    @property
    def Polygon_Count(self) -> int:
        return int(self.verts.shape[0])

    def Normal(self, Poly_ID: int):
        return tuple(self.normals[Poly_ID])

    def __getitem__(self, key):
        poly, corner = key
        return tuple(self.verts[poly, corner])

    def Polygon_Vertices(self, Poly_ID: int):
        return [tuple(self.verts[Poly_ID, c]) for c in range(int(self.nverts[Poly_ID]))]

    def _desc(self) -> capi.TopologyDesc:
        d = capi.TopologyDesc()
        d.P = self.Polygon_Count
        d.verts = ptr(self.verts)
        d.nverts = ptr(self.nverts)
        d.normals = ptr(self.normals)
        for a in range(3):
            d.min[a] = self.Min[a]
            d.max[a] = self.Max[a]
        return d


def _result_array(out, shape, dtype):
    """The result array of a host-buffer call: a fresh one, or the caller's `out` -- checked, never converted: the library writes
    straight into it.  Reusing one array across calls saves its first-touch page faults (56 B per ray: 1M rays 13.6 ms -> 1.9 ms
    per hare_shoot_batch_sharded call on an MI355X host, round 6)."""
    if out is None:
        return np.zeros(shape, dtype)
    shape = (shape,) if isinstance(shape, (int, np.integer)) else tuple(shape)
    if not (isinstance(out, np.ndarray) and out.dtype == dtype and out.shape == shape and out.flags.c_contiguous and out.flags.writeable):
        raise ValueError("out must be a writeable C-contiguous array of %d-byte result records with shape %s" % (np.dtype(dtype).itemsize, shape))
    return out


class Spatial_Partition:
    """Hare.Geometry.Spatial_Partition (Spatial_Partition.cs:27-35) over a native scene."""

    _kind = -1

    #: Opt-in (default off): reproduce the reference's `Ray_ID == 0` rule.  Voxel_Grid.Shoot and KDTree.Shoot skip a polygon whose
    #: mailbox entry equals R.Ray_ID (Voxel_Grid.cs:687-689, KDTree.cs:224-229) and the mailbox starts out all zero, so a ray with
    #: Ray_ID == 0 finds every polygon "already tested" and the reference returns X_Event() -- after moving an outside origin as for
    #: any ray.  The GPU classes keep no mailbox and return the hit; with this switch on, Shoot(R) with
    #: R.Ray_ID == 0 and Shoot_batch(..., ray_ids=) entries equal to 0 return the miss record on Voxel_Grid and KDTree.  (The
    #: reference's rule is stateful -- a polygon some OTHER ray of the same ThreadID tested since is no longer skipped; the switch
    #: reproduces the fresh-mailbox case, which is the one a caller that forgot to assign ids meets.)  Octree has no mailbox.
    mailbox_ray_id0 = False

    def __init__(self, Model_in: Sequence[Topology], device: int = 0):
        self.Model = list(Model_in)
        self.Char_Step = 0.0
        descs = (capi.TopologyDesc * len(self.Model))(*[t._desc() for t in self.Model])
        h = C.c_void_p()
        check(lib.hare_scene_create(descs, len(self.Model), int(device), C.byref(h)))
        self._h = h
        self.device = int(device)

    def set_option(self, name: str, value: int):
        """Diagnostics / A-B switch of this scene (hare_scene_set_option): e.g. ("voxel_kernel", 2) forces the pool kernel."""
        check(lib.hare_scene_set_option(self._h, name.encode(), int(value)))
        return self

    def get_option(self, name: str) -> int:
        """hare_scene_get_option: an option read back, or "voxel_tight_bytes" / "octree_scratch_bytes" (device memory of the accelerators)."""
        v = C.c_int64()
        check(lib.hare_scene_get_option(self._h, name.encode(), C.byref(v)))
        return int(v.value)

    def close(self):
        if getattr(self, "_h", None):
            lib.hare_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # bool Shoot(Ray R, int top_index, out X_Event Ret_event[, int poly_origin1, int poly_origin2 = -1])
    def Shoot(self, R: Ray, top_index: int, poly_origin1: int = -1, poly_origin2: int = -1):
        """One ray, like the reference call site: runs on the calling host thread (hare_shoot_one -- a GPU round trip
        per ray would be ~100x slower than the reference), bit-identical to the batch kernels.  Returns (Hit, X_Event)."""
        ray = np.array([R.x, R.y, R.z, R.dx, R.dy, R.dz], np.float64)
        ev = np.zeros(1, XEVENT_DTYPE)
        check(lib.hare_shoot_one(self._h, self._kind, int(top_index), ptr(ray), int(poly_origin1), int(poly_origin2), ptr(ev)))
        R.x, R.y, R.z = (float(c) for c in ray[:3])       # the reference moves R when it starts outside (F11)
        if self.mailbox_ray_id0 and self._kind != KIND_OCTREE and int(getattr(R, "Ray_ID", 1)) == 0:
            return False, X_Event()                      # Voxel_Grid.cs:687-689 / KDTree.cs:224-229 on a fresh mailbox
        e = X_Event.from_record(ev[0])
        return e.Hit, e

    def Shoot_one(self, ray6, top_index: int = 0, poly_origin1: int = -1, poly_origin2: int = -1):
        """hare_shoot_one on a raw [x,y,z,dx,dy,dz] array (updated in place like the reference moves R); returns the record."""
        ev = np.zeros(1, XEVENT_DTYPE)
        check(lib.hare_shoot_one(self._h, self._kind, int(top_index), ptr(ray6), int(poly_origin1), int(poly_origin2), ptr(ev)))
        return ev[0]

    def Occluded_batch(self, rays, t_max=None, top_index: int = 0, poly_origin1=None, poly_origin2=None, events: bool = True,
                       simple_kernel: bool = False):
        """Harness-defined occlusion predicate: closest hit exists and t < t_max (t_max None: any hit).
        Returns (occluded int32[n], events) -- or, with events=False, (occluded, counters): flags only, from the kernels that
        end a ray's traversal as soon as its flag is decided (hits = number of occluded rays)."""
        rays = np.array(rays, np.float64, order="C").reshape(-1, 6)
        n = rays.shape[0]
        occ = np.zeros(n, np.int32)
        out = np.zeros(n, XEVENT_DTYPE) if events else None
        tm = None if t_max is None else np.ascontiguousarray(np.broadcast_to(np.asarray(t_max, np.float64), (n,)))
        e1 = None if poly_origin1 is None else np.ascontiguousarray(poly_origin1, np.int32)
        e2 = None if poly_origin2 is None else np.ascontiguousarray(poly_origin2, np.int32)
        ctr = capi.Counters()
        check(lib.hare_occluded_batch(self._h, self._kind, int(top_index), n, ptr(rays), ptr(e1), ptr(e2), ptr(tm),
                                      capi.SHOOT_SIMPLE_KERNEL if simple_kernel else 0, ptr(occ), ptr(out), C.addressof(ctr)))
        return (occ, out) if events else (occ, ctr.as_dict())

    def occluded_device(self, n: int, d_rays: int, d_events: int, d_occluded: int, d_tmax: int = 0, top_index: int = 0,
                        d_excl1: int = 0, d_excl2: int = 0, d_counters: int = 0, stream: int = 0, flags: int = 0):
        check(lib.hare_occluded_device(self._h, self._kind, int(top_index), int(n), d_rays or None, d_excl1 or None,
                                       d_excl2 or None, d_tmax or None, int(flags), d_events or None, d_occluded or None,
                                       d_counters or None, stream or None))

    def Shoot_batch(self, rays, top_index: int = 0, poly_origin1=None, poly_origin2=None,
                    writeback_origin: bool = False, count_work: bool = False, simple_kernel: bool = False, slim: bool = False,
                    ray_ids=None, out=None):
        """n rays [n,6] through the HIP kernel (host buffers).  Returns (events, counters dict).
        With writeback_origin the rays array is updated in place like the reference mutates R.
        slim=True: the events come back as slim records (capi.SLIM_DTYPE for Voxel_Grid, SLIM_UV_DTYPE for the trees; 16 / 32
        bytes over the host link instead of 56); expand_events(rays, records) rebuilds the X_Events bit for bit.
        ray_ids (optional, one Ray_ID per ray) only matters with `mailbox_ray_id0` on: entries equal to 0 come back as miss records.
        out (optional): the caller's result array [n] of the dtype the call returns, written in place and returned (a caller that
        keeps it across calls does not pay a fresh array's page faults)."""
        if writeback_origin and not (isinstance(rays, np.ndarray) and rays.dtype == np.float64 and rays.flags.c_contiguous):
            rays = np.array(rays, np.float64, order="C")              # nothing of the caller's to write back into
        else:
            rays = np.ascontiguousarray(rays, np.float64)             # without the flag the library only reads them: no copy of 48 B per ray
        rays = rays.reshape(-1, 6)
        n = rays.shape[0]
        out = _result_array(out, n, self._slim_dtype() if slim else XEVENT_DTYPE)
        e1 = None if poly_origin1 is None else np.ascontiguousarray(poly_origin1, np.int32)
        e2 = None if poly_origin2 is None else np.ascontiguousarray(poly_origin2, np.int32)
        for e in (e1, e2):
            if e is not None and e.shape != (n,):
                raise ValueError("poly_origin arrays must have one entry per ray")
        flags = ((capi.SHOOT_WRITEBACK_ORIGIN if writeback_origin else 0) | (capi.SHOOT_COUNT_WORK if count_work else 0)
                 | (capi.SHOOT_SIMPLE_KERNEL if simple_kernel else 0) | (capi.SHOOT_SLIM_EVENTS if slim else 0))
        ctr = capi.Counters()
        check(lib.hare_shoot_batch(self._h, self._kind, int(top_index), n, ptr(rays), ptr(e1), ptr(e2), flags,
                                   ptr(out), C.addressof(ctr)))
        ctr = ctr.as_dict()
        if self.mailbox_ray_id0 and ray_ids is not None and self._kind != KIND_OCTREE:
            zero = np.asarray(ray_ids).reshape(-1) == 0
            if zero.shape != (n,):
                raise ValueError("ray_ids must have one entry per ray")
            if zero.any():
                ctr["hits"] -= int(np.count_nonzero(out["hit"][zero]))
                miss = np.zeros(1, out.dtype)
                miss["poly_id"] = -1
                out[zero] = miss[0]
        return out, ctr

    def _slim_dtype(self):
        return capi.SLIM_DTYPE if self._kind == KIND_VOXEL else capi.SLIM_UV_DTYPE

    def expand_events(self, rays, slim_records):
        """hare_expand_events: slim records (of a Shoot_batch(..., slim=True) on these rays, as they were passed in) -> X_Events."""
        rays = np.ascontiguousarray(rays, np.float64).reshape(-1, 6)
        rec = np.ascontiguousarray(slim_records, self._slim_dtype())
        out = np.zeros(len(rec), XEVENT_DTYPE)
        check(lib.hare_expand_events(self._h, self._kind, len(rec), ptr(rays), ptr(rec), ptr(out)))
        return out

    @staticmethod
    def Shoot_batch_sharded(partitions, rays, top_index: int = 0, poly_origin1=None, poly_origin2=None,
                            writeback_origin: bool = False, slim: bool = False, out=None):
        """One batch over several devices from one process: `partitions` are equal partitions built on different
        devices (e.g. [Voxel_Grid(model, 64, device=k) for k in range(G)]); rays are split into contiguous shards in
        that order (hare_shoot_batch_sharded).  Returns (events, summed counters), byte-identical to Shoot_batch.
        out (optional): the caller's result array, as in Shoot_batch."""
        parts = list(partitions)
        if not parts or any(p._kind != parts[0]._kind for p in parts):
            raise ValueError("need one or more partitions of the same kind")
        if writeback_origin and not (isinstance(rays, np.ndarray) and rays.dtype == np.float64 and rays.flags.c_contiguous):
            rays = np.array(rays, np.float64, order="C")              # nothing of the caller's to write back into
        else:
            rays = np.ascontiguousarray(rays, np.float64)             # without the flag the library only reads them: no copy of 48 B per ray
        rays = rays.reshape(-1, 6)
        n = rays.shape[0]
        out = _result_array(out, n, parts[0]._slim_dtype() if slim else XEVENT_DTYPE)
        e1 = None if poly_origin1 is None else np.ascontiguousarray(poly_origin1, np.int32)
        e2 = None if poly_origin2 is None else np.ascontiguousarray(poly_origin2, np.int32)
        for e in (e1, e2):
            if e is not None and e.shape != (n,):
                raise ValueError("poly_origin arrays must have one entry per ray")
        handles = (C.c_void_p * len(parts))(*[p._h for p in parts])
        ctr = capi.Counters()
        check(lib.hare_shoot_batch_sharded(handles, len(parts), parts[0]._kind, int(top_index), n, ptr(rays), ptr(e1), ptr(e2),
                                           (capi.SHOOT_WRITEBACK_ORIGIN if writeback_origin else 0) | (capi.SHOOT_SLIM_EVENTS if slim else 0),
                                           ptr(out), C.addressof(ctr)))
        return out, ctr.as_dict()

    def Bounce_batch(self, rays, bounces: int, top_index: int = 0, poly_origin1=None, poly_origin2=None, all_casts: bool = False,
                     per_cast: bool = False, simple_kernel: bool = False, out=None):
        """The device-resident specular bounce loop from host buffers (hare_bounce_batch): `bounces` casts with a reflection
        between them, rays resident on the GPU throughout.  Returns (events, counters) -- events of the LAST cast [n], or with
        all_casts=True of every cast [bounces, n]; with per_cast=True a third value: the list of per-cast counter dicts."""
        rays = np.ascontiguousarray(rays, np.float64).reshape(-1, 6)
        n, B = rays.shape[0], int(bounces)
        e1 = None if poly_origin1 is None else np.ascontiguousarray(poly_origin1, np.int32)
        e2 = None if poly_origin2 is None else np.ascontiguousarray(poly_origin2, np.int32)
        ev_all = _result_array(out, (B, n), XEVENT_DTYPE) if all_casts else None          # out: the caller's array, as in Shoot_batch
        ev_last = None if all_casts else _result_array(out, n, XEVENT_DTYPE)
        ctr = capi.Counters()
        pcs = (capi.Counters * max(B, 1))()
        check(lib.hare_bounce_batch(self._h, self._kind, int(top_index), n, ptr(rays), ptr(e1), ptr(e2), B,
                                    capi.SHOOT_SIMPLE_KERNEL if simple_kernel else 0, ptr(ev_all), ptr(ev_last), C.addressof(ctr),
                                    C.addressof(pcs)))
        out = (ev_all if all_casts else ev_last, ctr.as_dict())
        return out + ([pcs[b].as_dict() for b in range(B)],) if per_cast else out

    @staticmethod
    def Bounce_batch_sharded(partitions, rays, bounces: int, top_index: int = 0, all_casts: bool = False, out=None):
        """hare_bounce_batch_sharded: the loop over several devices from one process (contiguous ray shards, as Shoot_batch_sharded)."""
        parts = list(partitions)
        if not parts or any(p._kind != parts[0]._kind for p in parts):
            raise ValueError("need one or more partitions of the same kind")
        rays = np.ascontiguousarray(rays, np.float64).reshape(-1, 6)
        n, B = rays.shape[0], int(bounces)
        ev_all = _result_array(out, (B, n), XEVENT_DTYPE) if all_casts else None          # out: the caller's array, as in Shoot_batch
        ev_last = None if all_casts else _result_array(out, n, XEVENT_DTYPE)
        handles = (C.c_void_p * len(parts))(*[p._h for p in parts])
        ctr = capi.Counters()
        check(lib.hare_bounce_batch_sharded(handles, len(parts), parts[0]._kind, int(top_index), n, ptr(rays), None, None, B, 0,
                                            ptr(ev_all), ptr(ev_last), C.addressof(ctr), None))
        return (ev_all if all_casts else ev_last), ctr.as_dict()

    def shoot_device(self, n: int, d_rays: int, d_out: int, top_index: int = 0, d_excl1: int = 0, d_excl2: int = 0,
                     d_counters: int = 0, stream: int = 0, flags: int = 0):
        """Device-resident shoot: raw device addresses (e.g. torch.Tensor.data_ptr()) + a hipStream_t."""
        check(lib.hare_shoot_device(self._h, self._kind, int(top_index), int(n), d_rays or None, d_excl1 or None,
                                    d_excl2 or None, int(flags), d_out or None, d_counters or None, stream or None))

    def kernel_name(self, n: int, top_index: int = 0, flags: int = 0) -> str:
        """The gfx950 kernel a shoot of n rays launches (what rocprofv3 will list)."""
        return (lib.hare_shoot_kernel_name(self._h, self._kind, int(top_index), int(n), int(flags)) or b"").decode()

    def bounce_kernel_name(self, n: int, bounces: int, top_index: int = 0) -> str:
        """The fused kernel hare_bounce_device launches for n rays, or "" when it runs a launch per cast."""
        if bounces > 16:
            return ""
        name = self.kernel_name(n, top_index, flags=32)          # HARE_SHOOT_BOUNCE_LOOP
        return name if "bounce" in name else ""

    def reflect_device(self, n: int, d_rays: int, d_events: int, d_excl_out: int, top_index: int = 0, stream: int = 0):
        check(lib.hare_reflect_device(self._h, int(top_index), int(n), d_rays, d_events, d_excl_out, stream or None))

    def bounce_device(self, n: int, d_rays: int, bounces: int, d_work: int, d_events_last: int = 0, d_events_all: int = 0,
                      top_index: int = 0, d_excl1: int = 0, d_excl2: int = 0, d_counters: int = 0, d_counters_per_cast: int = 0,
                      stream: int = 0, flags: int = 0):
        """The bounce loop on device buffers (hare_bounce_device): `bounces` casts per ray, reflection + exclusion of the polygon left
        between them; d_rays is read and overwritten, d_work is 2 n int32 of scratch.  One launch for a Voxel_Grid where the pool
        kernel serves, else a launch per cast; stream-ordered, no host synchronisation."""
        check(lib.hare_bounce_device(self._h, self._kind, int(top_index), int(n), d_rays or None, d_excl1 or None, d_excl2 or None,
                                     int(bounces), int(flags), d_work or None, d_events_all or None, d_events_last or None,
                                     d_counters or None, d_counters_per_cast or None, stream or None))


class Voxel_Grid(Spatial_Partition):
    """Voxel_Grid(Topology[] Model_in, int Domain) / (Topology[] Model_in, int MaxDomain, int Avg_polys)
    (Voxel_Grid.cs:48, :128)."""

    _kind = KIND_VOXEL

    def __init__(self, Model_in, Domain: int, Avg_polys: Optional[int] = None, device: int = 0):
        super().__init__(Model_in, device)
        if Avg_polys is None:
            check(lib.hare_voxel_build(self._h, int(Domain)))
        else:
            check(lib.hare_voxel_build_adaptive(self._h, int(Domain), int(Avg_polys)))
        info = self.info()
        self.Char_Step = info.char_step
        self.VoxelCt = info.ct

    def info(self) -> capi.VoxelInfo:
        i = capi.VoxelInfo()
        check(lib.hare_voxel_get_info(self._h, C.addressof(i)))
        return i

    # public members of the reference class
    @property
    def Xdim(self):
        return self.info().box_dims[0]

    @property
    def Ydim(self):
        return self.info().box_dims[1]

    @property
    def Zdim(self):
        return self.info().box_dims[2]

    @property
    def MinPt(self):
        return tuple(self.info().obox_min)

    def PointInVoxel(self, Pt):
        """Voxel_Grid.PointInVoxel(Point, out X, out Y, out Z) (Voxel_Grid.cs:322-327)."""
        i = self.info()
        return tuple(int(np.floor((Pt[a] - i.obox_min[a]) / i.voxel_dims[a])) for a in range(3))

    def PointInVoxel_code(self, Pt) -> int:
        """int Voxel_Grid.PointInVoxel(Point) (Voxel_Grid.cs:329-332): the VoxelCode of the voxel holding Pt."""
        return self.VoxelCode(*self.PointInVoxel(Pt))

    def VoxelDecode(self, Code: int):
        """Voxel_Grid.VoxelDecode (Voxel_Grid.cs:256-262) -> (X, Y, Z)."""
        ct = self.VoxelCt
        Z = Code // (ct * ct)
        Code -= Z * ct * ct
        Y = Code // ct
        return Code - Y * ct, Y, Z

    def VoxelCode(self, X: int, Y: int, Z: int) -> int:
        """Voxel_Grid.VoxelCode (Voxel_Grid.cs:264-267): XYTot * Z + VoxelCtY * X + Y."""
        ct = self.VoxelCt
        return ct * ct * Z + ct * X + Y

    def Voxel_Inv(self, top_index: int = 0):
        """Voxel_Inv[x,y,z,top] as CSR (cell = (x*ct + y)*ct + z)."""
        i = self.info()
        n = i.ct ** 3
        start = np.zeros(n + 1, np.uint32)
        check(lib.hare_voxel_get_lists(self._h, int(top_index), ptr(start), None))
        items = np.zeros(max(1, int(start[-1])), np.int32)
        check(lib.hare_voxel_get_lists(self._h, int(top_index), ptr(start), ptr(items)))
        return start, items[: int(start[-1])]


class Octree(Spatial_Partition):
    """Octree(Topology[] Model_In, int maxDepth, int maxPolygonsPerNode) ("Octree - alt.cs":45)."""

    _kind = KIND_OCTREE

    def __init__(self, Model_In, maxDepth: int, maxPolygonsPerNode: int, device: int = 0):
        super().__init__(Model_In, device)
        check(lib.hare_octree_build(self._h, int(maxDepth), int(maxPolygonsPerNode)))

    def info(self) -> capi.TreeInfo:
        i = capi.TreeInfo()
        check(lib.hare_octree_get_info(self._h, C.addressof(i)))
        return i

    def nodes(self):
        i = self.info()
        n, tot = i.n_nodes, int(i.total_items)
        boxes = np.zeros((n, 6))
        fc = np.zeros(n, np.int32)
        st = np.zeros(n, np.int32)
        cn = np.zeros(n, np.int32)
        items = np.zeros(max(tot, 1), np.int32)
        check(lib.hare_octree_get_nodes(self._h, ptr(boxes), ptr(fc), ptr(st), ptr(cn), ptr(items)))
        return boxes, fc, st, cn, items[:tot]


class KDTree(Spatial_Partition):
    """KDTree(Topology[] Model_In, int maxDepth, int maxPolygonsPerNode) (KDTree.cs:51)."""

    _kind = KIND_KDTREE

    def __init__(self, Model_In, maxDepth: int, maxPolygonsPerNode: int, device: int = 0):
        super().__init__(Model_In, device)
        check(lib.hare_kdtree_build(self._h, int(maxDepth), int(maxPolygonsPerNode)))

    def info(self) -> capi.TreeInfo:
        i = capi.TreeInfo()
        check(lib.hare_kdtree_get_info(self._h, C.addressof(i)))
        return i

    def nodes(self):
        i = self.info()
        n, tot = i.n_nodes, int(i.total_items)
        boxes = np.zeros((n, 6))
        split = np.zeros(n)
        axis = np.zeros(n, np.int32)
        left = np.zeros(n, np.int32)
        right = np.zeros(n, np.int32)
        st = np.zeros(n, np.int32)
        cn = np.zeros(n, np.int32)
        items = np.zeros(max(tot, 1), np.int32)
        check(lib.hare_kdtree_get_nodes(self._h, ptr(boxes), ptr(split), ptr(axis), ptr(left), ptr(right), ptr(st),
                                        ptr(cn), ptr(items)))
        return boxes, split, axis, left, right, st, cn, items[:tot]
