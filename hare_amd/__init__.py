"""hare_amd -- MI355X (gfx950) implementation of Hare's ray-cast hot path, Spatial_Partition.Shoot.

The package is a thin host-side mirror of the reference interface (geometry.py) over the C-ABI
library libhare_hip.so (csrc/, include/hare_hip.h) plus the synthetic scene/ray generators the
harness uses (scenes.py).  Importing it loads the library and fails loudly if it is missing:
there is no CPU or Python fallback for the path.
"""
from . import capi  # noqa: F401  (raises ImportError when libhare_hip.so is absent)
from .capi import HareError, device_count  # noqa: F401
from .geometry import KDTree, Octree, Ray, Spatial_Partition, Topology, Voxel_Grid, X_Event  # noqa: F401
from . import scenes  # noqa: F401

__all__ = ["Ray", "X_Event", "Topology", "Spatial_Partition", "Voxel_Grid", "Octree", "KDTree",
           "HareError", "device_count", "scenes", "capi"]
