"""The C-ABI library loads on a GPU-less box and exports every symbol include/hare_hip.h declares;
wire structs have the documented sizes; argument errors follow the ABI's error convention.
No compute call is made here (those are the -m gpu tests)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import hare_amd as H
from hare_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "hare_hip.h")).read()
    return sorted(set(re.findall(r"HARE_API[^;(]*?\b(hare_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    names = declared_symbols()
    assert len(names) >= 20
    raw = C.CDLL(capi.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"libhare_hip.so does not export {n}"
    assert sorted(capi.SYMBOLS) == names, "hare_amd/capi.py and include/hare_hip.h disagree"


def test_header_compiles_as_c_and_struct_sizes(tmp_path):
    src = tmp_path / "t.c"
    src.write_text(
        '#include "hare_hip.h"\n#include <stdio.h>\n'
        "int main(void){printf(\"%zu %zu %zu %zu\\n\", sizeof(hare_ray), sizeof(hare_xevent),"
        " sizeof(hare_counters), sizeof(hare_topology_desc)); return 0;}\n")
    exe = tmp_path / "t"
    import subprocess
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).split()
    assert [int(x) for x in out] == [48, 56, 64, C.sizeof(capi.TopologyDesc)]
    assert capi.RAY_DTYPE.itemsize == 48 and capi.XEVENT_DTYPE.itemsize == 56 and C.sizeof(capi.Counters) == 64


def test_library_does_not_link_a_hip_runtime_or_the_oracle():
    import subprocess
    needed = subprocess.check_output(["readelf", "-d", capi.LIB_PATH]).decode()
    libs = re.findall(r"NEEDED.*\[(.*?)\]", needed)
    assert not any("amdhip" in l or "oracle" in l or "torch" in l for l in libs), libs


def test_argument_errors_use_error_codes_not_exceptions():
    h = C.c_void_p()
    assert capi.lib.hare_scene_create(None, 1, 0, C.byref(h)) == capi.HARE_E_INVALID
    assert "hare_scene_create" in capi.last_error()
    assert capi.lib.hare_voxel_build(None, 8) == capi.HARE_E_INVALID
    # a pentagon is refused like Topology does (Hare_Geometry_Topology.cs:298)
    v = np.zeros((1, 4, 3))
    with pytest.raises(H.HareError) as ei:
        H.Voxel_Grid([H.Topology(v, np.array([5], np.int32), normals=np.zeros((1, 3)), Min=np.zeros(3), Max=np.ones(3))], 4)
    assert ei.value.code == capi.HARE_E_UNSUPPORTED and "more than 4 sides" in str(ei.value)
    m = H.scenes.shoebox()
    with pytest.raises(H.HareError) as ei:
        H.Voxel_Grid([H.Topology(m.verts, m.nverts)], 0)
    assert ei.value.code == capi.HARE_E_INVALID


def test_shoot_without_partition_or_gpu_fails_loudly(gpu_available):
    m = H.scenes.shoebox()
    part = H.Spatial_Partition([H.Topology(m.verts, m.nverts)])
    part._kind = capi.KIND_VOXEL
    rays = H.scenes.random_rays(4, m.size)
    with pytest.raises(H.HareError) as ei:
        part.Shoot_batch(rays)
    # no GPU -> HARE_E_NODEVICE; GPU but nothing built -> HARE_E_STATE.  Never a silent CPU answer.
    assert ei.value.code == (capi.HARE_E_STATE if gpu_available else capi.HARE_E_NODEVICE)


def test_slim_events_with_origin_writeback_is_refused_before_anything_runs():
    """HARE_SHOOT_SLIM_EVENTS | HARE_SHOOT_WRITEBACK_ORIGIN: the write-back would overwrite the caller's rays with the moved origins
    and hare_expand_events would then rebuild t without t_start for every ray that started outside the grid -- silently.  The
    call is refused (HARE_E_INVALID) before any device work, so this needs no GPU; the rays are untouched."""
    m = H.scenes.shoebox()
    g = H.Voxel_Grid([H.Topology(m.verts, m.nverts)], 8)
    rays = H.scenes.burst_rays(64, m.size)
    rays[:, :3] += 100.0                     # origins outside the grid: exactly the rays the combination would get wrong
    before = rays.copy()
    with pytest.raises(H.HareError) as ei:
        g.Shoot_batch(rays, writeback_origin=True, slim=True)
    assert ei.value.code == capi.HARE_E_INVALID and "WRITEBACK_ORIGIN" in str(ei.value)
    assert np.array_equal(rays, before)
    out = np.zeros(64, capi.SLIM_DTYPE)
    rc = capi.lib.hare_shoot_batch(g._h, capi.KIND_VOXEL, 0, 64, rays.ctypes.data, None, None,
                                   capi.SHOOT_SLIM_EVENTS | capi.SHOOT_WRITEBACK_ORIGIN, out.ctypes.data, None)
    assert rc == capi.HARE_E_INVALID


def test_bounce_device_without_a_gpu_fails_loudly_and_the_kernel_name_query_follows_the_option():
    """hare_bounce_device has no CPU fallback (HARE_E_NODEVICE on a GPU-less box); hare_shoot_kernel_name with HARE_SHOOT_BOUNCE_LOOP names
    the one-launch kernel only when the scene option asks for it (the default is a launch per cast)."""
    m = H.scenes.shoebox()
    g = H.Voxel_Grid([H.Topology(m.verts, m.nverts)], 8)
    n = 1 << 20
    assert g.kernel_name(n) == "hare_voxel_pool_tri"
    assert g.bounce_kernel_name(n, 8) == ""
    g.set_option("bounce_fused", 1)
    assert g.bounce_kernel_name(n, 8) == "hare_voxel_bounce_tri" and g.bounce_kernel_name(n, 17) == ""
    assert g.kernel_name(n, flags=capi.SHOOT_BOUNCE_LOOP | capi.SHOOT_SIMPLE_KERNEL) == "hare_voxel_shoot_tri"
    g.set_option("bounce_fused", 0)
    if H.device_count() == 0:
        rc = capi.lib.hare_bounce_device(g._h, capi.KIND_VOXEL, 0, 16, 1, None, None, 4, 0, 1, None, 1, None, None, None)
        assert rc == capi.HARE_E_NODEVICE
    o = H.Octree([H.Topology(m.verts, m.nverts)], 4, 8)
    assert o.kernel_name(1000) == "hare_octree_group" and o.kernel_name(1 << 20) == "hare_octree_dense"        # the crossover rule, 256-CU part
    o.set_option("octree_kernel", 1)
    assert o.kernel_name(1000) == "hare_octree_persist"


def test_the_kernel_rule_for_octree_batches_and_the_tight_box_options():
    """The K2g / K2d crossover is 320 rays per CU (81 920 rays on the 256-CU part the rule assumes without a device; round 6: it was 768); the
    tight-box switches are scene options with values 0 / 1 (anything else is refused), and an unknown name is an error, not a no-op."""
    m = H.scenes.shoebox()
    o = H.Octree([H.Topology(m.verts, m.nverts)], 4, 8)
    assert o.kernel_name(81_919) == "hare_octree_group" and o.kernel_name(81_920) == "hare_octree_dense"
    for name in ("octree_tight", "voxel_tight"):
        o.set_option(name, 0)
        o.set_option(name, 1)
        with pytest.raises(H.HareError):
            o.set_option(name, 2)
    with pytest.raises(H.HareError):
        o.set_option("octree_tights", 1)


def test_the_callers_result_array_is_checked_never_converted():
    """`out=` of the host-buffer calls of the Python mirror (Shoot_batch, Shoot_batch_sharded, Bounce_batch, Bounce_batch_sharded): the
    library writes straight into the caller's array, so a wrong length, record type, layout or a read-only array is a ValueError before
    anything is called -- never a silent copy.  (What it is for: a caller that keeps its result array across calls does not pay a fresh
    array's page faults, tests/test_gpu_round6.py.)  Needs no GPU: a good array gets as far as the library, which has no CPU path."""
    m = H.scenes.shoebox()
    g = H.Voxel_Grid([H.Topology(m.verts, m.nverts)], 8)
    rays = H.scenes.burst_rays(100, m.size)
    ro = np.zeros(100, capi.XEVENT_DTYPE)
    ro.flags.writeable = False
    bad = (np.zeros(99, capi.XEVENT_DTYPE), np.zeros(100, capi.SLIM_DTYPE), np.zeros((100, 7)), np.zeros(200, capi.XEVENT_DTYPE)[::2], ro, [0] * 100)
    SP = H.Spatial_Partition
    for b in bad:
        for call in (lambda: g.Shoot_batch(rays, out=b), lambda: SP.Shoot_batch_sharded([g], rays, out=b),
                     lambda: g.Bounce_batch(rays, 3, out=b), lambda: SP.Bounce_batch_sharded([g], rays, 3, out=b)):
            with pytest.raises(ValueError, match="out must be"):
                call()
    with pytest.raises(ValueError, match="out must be"):
        g.Bounce_batch(rays, 3, all_casts=True, out=np.zeros(100, capi.XEVENT_DTYPE))          # all casts: [bounces, n]
    with pytest.raises(ValueError, match="out must be"):
        g.Shoot_batch(rays, slim=True, out=np.zeros(100, capi.XEVENT_DTYPE))                   # slim records are another type
    if H.device_count() == 0:
        for call in (lambda: g.Shoot_batch(rays, out=np.zeros(100, capi.XEVENT_DTYPE)), lambda: g.Shoot_batch(rays, slim=True, out=np.zeros(100, capi.SLIM_DTYPE)),
                     lambda: g.Bounce_batch(rays, 3, all_casts=True, out=np.zeros((3, 100), capi.XEVENT_DTYPE))):
            with pytest.raises(H.HareError) as ei:
                call()
            assert ei.value.code == capi.HARE_E_NODEVICE
