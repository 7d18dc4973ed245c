"""Grid build time: GPU builder vs host builder (1M-tri cathedral, D=128; hall D=64)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hare_amd as H
for scene, D in (("hall", 64), ("cathedral", 128)):
    m = H.scenes.SCENES[scene](); T = H.Topology(m.verts, m.nverts)
    for mode in ("gpu", "host"):
        if mode == "host": os.environ["HARE_BUILD"] = "host"
        else: os.environ.pop("HARE_BUILD", None)
        H.Voxel_Grid([T], D)
        t0 = time.perf_counter(); g = H.Voxel_Grid([T], D); dt = time.perf_counter() - t0
        print(scene, m.P, "tris D=%d" % D, mode, "build+upload %.1f ms" % (dt * 1e3), "on_device", g.info().built_on_device, "items", g.info().total_items)
for scene, depth, polys in (("hall", 8, 16), ("cathedral", 8, 16)):
    m = H.scenes.SCENES[scene](); T = H.Topology(m.verts, m.nverts)
    for mode in ("gpu", "host"):
        if mode == "host": os.environ["HARE_BUILD"] = "host"
        else: os.environ.pop("HARE_BUILD", None)
        if mode == "gpu": H.Octree([T], depth, polys)          # warm-up (module load, allocator)
        t0 = time.perf_counter(); o = H.Octree([T], depth, polys); dt = time.perf_counter() - t0
        i = o.info()
        print(scene, m.P, "tris octree %d/%d" % (depth, polys), mode, "build+upload %.1f ms" % (dt * 1e3), "on_device", i.built_on_device,
              "nodes", i.n_nodes, "items", i.total_items, flush=True)
