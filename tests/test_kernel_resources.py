"""The production kernels keep their working set in registers: no VGPR of K1q (`hare_voxel_pool_*`) or K2d (`hare_octree_dense`) is spilled
and neither touches scratch (profiles/r06_scratch_sites.md) -- read from the metadata the compiler writes next to
the code object (hare_amd/csrc/build/hare_kernels.s, made by the library's Makefile).  Also the register budgets the launch geometry
assumes: three waves per SIMD = at most 168 VGPRs (launch.h: HARE_K2D_WAVES_PER_EU; voxel_pool.hip: twelve waves per CU)."""
import os
import re

import pytest

ASM = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hare_amd", "csrc", "build", "hare_kernels.s")


def kernels():
    txt = open(ASM).read()
    meta = txt[txt.index("amdhsa.kernels:"):]
    out = {}
    for blk in re.split(r"\n  - \.agpr_count:", meta)[1:]:
        def g(key):
            m = re.search(r"\." + key + r":\s+(\S+)", blk)
            return m.group(1) if m else None
        out[g("name")] = {k: int(g(k)) for k in ("vgpr_count", "vgpr_spill_count", "private_segment_fixed_size")}
    return out


@pytest.mark.skipif(not os.path.exists(ASM), reason="the library was not built here (no hare_kernels.s)")
def test_production_kernels_spill_nothing():
    k = kernels()
    for name in ("hare_voxel_pool_tri", "hare_voxel_pool_quad", "hare_voxel_pool_tri_g", "hare_voxel_pool_quad_g", "hare_octree_dense", "hare_octree_occl",
                 "hare_octree_group", "hare_kdtree_occl"):
        assert name in k, name
        assert k[name]["vgpr_spill_count"] == 0 and k[name]["private_segment_fixed_size"] == 0, (name, k[name])
    for name in ("hare_voxel_pool_tri", "hare_voxel_pool_quad_g", "hare_octree_dense", "hare_octree_dense_own", "hare_octree_occl"):
        assert k[name]["vgpr_count"] <= 168, (name, k[name])               # three waves per SIMD
    for name in ("hare_kdtree_dense", "hare_octree_group", "hare_octree_persist"):
        assert k[name]["vgpr_count"] <= 128, (name, k[name])               # four
