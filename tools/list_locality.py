"""Developer analysis (CPU, no GPU): how local are the polygon ids inside one cell's list?  The numbers behind the dense
pre-cull array (hare_device.h, kCullStride).  Uses the oracle's grid builder for the lists -- analysis only, not product.

    python tools/list_locality.py            # cathedral D=128 and hall D=64
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po
import hare_amd as H

for scene, D in (("cathedral", 128), ("hall", 64)):
    mesh = H.scenes.SCENES[scene]()
    off, items = (np.asarray(a) for a in po.VoxelGrid([po.Topology(mesh.verts, mesh.nverts)], domain=D).lists()[:2])
    cnt = np.diff(off)
    print("%s D=%d: %d polygons, %d non-empty cells, %d entries (%.1f per non-empty cell)" % (
        scene, D, len(mesh.nverts), int(np.count_nonzero(cnt)), len(items), cnt[cnt > 0].mean()))
    inside = np.ones(len(items) - 1, bool)                      # pairs (k, k+1) that lie in the same cell
    ends = off[1:-1]
    inside[ends[(ends > 0) & (ends < len(items))] - 1] = False
    d = np.abs(np.diff(items.astype(np.int64)))[inside]
    print("  neighbouring entries of a cell: ids differ by 1 in %.1f %%, median difference %d" % (100 * np.mean(d == 1), np.median(d)))
    cell_of = np.repeat(np.arange(len(cnt)), cnt)
    for stride, per_line in ((128, 1.0), (64, 2.0), (48, 128 / 48)):
        lines = np.unique(np.stack([cell_of, (items.astype(np.int64) * stride) >> 7], 1), axis=0).shape[0]
        print("  pre-cull records at a %3d-byte stride: %.2f distinct 128-byte lines per list entry" % (stride, lines / len(items)))
