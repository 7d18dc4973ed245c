"""ctypes binding of hare_amd/libhare_hip.so (the C-ABI declared in include/hare_hip.h).

The library is the product: if it is missing this module raises at import -- there is no Python
or CPU fallback for the ray-cast path.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HARE_LIB") or os.path.join(_HERE, "libhare_hip.so")   # HARE_LIB: developer A/B builds

HARE_OK = 0
HARE_E_INVALID, HARE_E_NOMEM, HARE_E_HIP, HARE_E_NODEVICE, HARE_E_STATE, HARE_E_UNSUPPORTED = -1, -2, -3, -4, -5, -6
KIND_VOXEL, KIND_OCTREE, KIND_KDTREE = 0, 1, 2
SHOOT_WRITEBACK_ORIGIN, SHOOT_COUNT_WORK, SHOOT_SIMPLE_KERNEL, SHOOT_RETIRED_RAYS, SHOOT_SLIM_EVENTS, SHOOT_BOUNCE_LOOP = 1, 2, 4, 8, 16, 32
SHOOT_COUNT_OWN = 64

RAY_DTYPE = np.dtype([("x", "<f8"), ("y", "<f8"), ("z", "<f8"), ("dx", "<f8"), ("dy", "<f8"), ("dz", "<f8")])
XEVENT_DTYPE = np.dtype(
    [("t", "<f8"), ("u", "<f8"), ("v", "<f8"), ("x", "<f8"), ("y", "<f8"), ("z", "<f8"),
     ("poly_id", "<i4"), ("hit", "<i4")]
)
SLIM_DTYPE = np.dtype([("t", "<f8"), ("poly_id", "<i4"), ("hit", "<i4")])                                  # hare_slim_event (Voxel_Grid)
SLIM_UV_DTYPE = np.dtype([("t", "<f8"), ("u", "<f8"), ("v", "<f8"), ("poly_id", "<i4"), ("hit", "<i4")])    # hare_slim_event_uv (trees)
assert RAY_DTYPE.itemsize == 48 and XEVENT_DTYPE.itemsize == 56 and SLIM_DTYPE.itemsize == 16 and SLIM_UV_DTYPE.itemsize == 32


class HareError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"hare_hip error {code}: {msg}")
        self.code = code


class Counters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("hits", C.c_uint64), ("cells", C.c_uint64), ("entries", C.c_uint64),
                ("tests", C.c_uint64), ("reserved", C.c_uint64 * 3)]

    def as_dict(self):
        d = {k: int(getattr(self, k)) for k in ("rays", "hits", "cells", "entries", "tests")}
        d["culls"] = int(self.reserved[0])       # HARE_SHOOT_COUNT_OWN: candidates pre-culled
        d["steps"] = int(self.reserved[1])       # HARE_SHOOT_COUNT_OWN, Voxel_Grid: walk operations executed (steps, or block jumps under "voxel_skip")
        return d


class TopologyDesc(C.Structure):
    _fields_ = [("P", C.c_int32), ("reserved", C.c_int32), ("verts", C.c_void_p), ("nverts", C.c_void_p),
                ("normals", C.c_void_p), ("min", C.c_double * 3), ("max", C.c_double * 3)]


class VoxelInfo(C.Structure):
    _fields_ = [("ct", C.c_int32), ("n_topos", C.c_int32), ("obox_min", C.c_double * 3), ("obox_max", C.c_double * 3),
                ("box_dims", C.c_double * 3), ("voxel_dims", C.c_double * 3), ("char_step", C.c_double),
                ("total_items", C.c_uint64), ("built_on_device", C.c_int32), ("reserved", C.c_int32)]


class TreeInfo(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("max_depth", C.c_int32), ("max_polys", C.c_int32), ("built_on_device", C.c_int32),
                ("total_items", C.c_uint64)]


# every symbol include/hare_hip.h declares: (name, restype, argtypes)
_vp, _i32, _i64, _u32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32
SYMBOLS = {
    "hare_version": (C.c_char_p, []),
    "hare_last_error": (C.c_char_p, []),
    "hare_device_count": (C.c_int, [_vp]),
    "hare_hip_runtime_path": (C.c_char_p, []),
    "hare_polygon_normals": (C.c_int, [_vp, _vp, _i32, _vp]),
    "hare_topology_bounds": (C.c_int, [_vp, _vp, _i32, _vp, _vp]),
    "hare_topology_ingest": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "hare_scene_create": (C.c_int, [_vp, _i32, _i32, _vp]),
    "hare_scene_destroy": (None, [_vp]),
    "hare_scene_set_option": (C.c_int, [_vp, C.c_char_p, _i64]),
    "hare_scene_get_option": (C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_int64)]),
    "hare_voxel_build": (C.c_int, [_vp, _i32]),
    "hare_voxel_build_adaptive": (C.c_int, [_vp, _i32, _i32]),
    "hare_octree_build": (C.c_int, [_vp, _i32, _i32]),
    "hare_kdtree_build": (C.c_int, [_vp, _i32, _i32]),
    "hare_voxel_get_info": (C.c_int, [_vp, _vp]),
    "hare_voxel_get_lists": (C.c_int, [_vp, _i32, _vp, _vp]),
    "hare_octree_get_info": (C.c_int, [_vp, _vp]),
    "hare_octree_get_nodes": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "hare_kdtree_get_info": (C.c_int, [_vp, _vp]),
    "hare_kdtree_get_nodes": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hare_shoot_batch": (C.c_int, [_vp, _i32, _i32, _i64, _vp, _vp, _vp, _u32, _vp, _vp]),
    "hare_expand_events": (C.c_int, [_vp, _i32, _i64, _vp, _vp, _vp]),
    "hare_shoot_batch_sharded": (C.c_int, [_vp, _i32, _i32, _i32, _i64, _vp, _vp, _vp, _u32, _vp, _vp]),
    "hare_shoot_device": (C.c_int, [_vp, _i32, _i32, _i64, _vp, _vp, _vp, _u32, _vp, _vp, _vp]),
    "hare_reflect_device": (C.c_int, [_vp, _i32, _i64, _vp, _vp, _vp, _vp]),
    "hare_bounce_device": (C.c_int, [_vp, _i32, _i32, _i64, _vp, _vp, _vp, _i32, _u32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hare_shoot_kernel_name": (C.c_char_p, [_vp, _i32, _i32, _i64, _u32]),
    "hare_shoot_one": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _i32, _vp]),
    "hare_bounce_batch": (C.c_int, [_vp, _i32, _i32, _i64, _vp, _vp, _vp, _i32, _u32, _vp, _vp, _vp, _vp]),
    "hare_bounce_batch_sharded": (C.c_int, [_vp, _i32, _i32, _i32, _i64, _vp, _vp, _vp, _i32, _u32, _vp, _vp, _vp, _vp]),
    "hare_occluded_device": (C.c_int, [_vp, _i32, _i32, _i64, _vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp]),
    "hare_occluded_batch": (C.c_int, [_vp, _i32, _i32, _i64, _vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp]),
    "hare_occluded_batch_sharded": (C.c_int, [_vp, _i32, _i32, _i32, _i64, _vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp]),
}

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(or `make -C hare_amd/csrc`). hare_amd has no CPU fallback."
    )

lib = C.CDLL(LIB_PATH)
for _name, (_res, _args) in SYMBOLS.items():
    _f = getattr(lib, _name)   # AttributeError here == the library does not export the ABI
    _f.restype = _res
    _f.argtypes = _args


def last_error() -> str:
    return (lib.hare_last_error() or b"").decode()


def check(rc: int) -> None:
    if rc != HARE_OK:
        raise HareError(rc, last_error())


def device_count() -> int:
    n = C.c_int32(0)
    rc = lib.hare_device_count(C.addressof(n))
    if rc == HARE_E_NODEVICE:
        return 0
    check(rc)
    return int(n.value)


def ptr(a):
    return None if a is None else a.ctypes.data
