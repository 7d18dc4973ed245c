"""Developer A/B for the octree kernels: K2p (hare_octree_persist) vs K2q (hare_octree_pool, octree_pool.hip), over several
builds of libhare_hip (HARE_LIB).  One subprocess per variant; a hung kernel stops the run.  Per variant: X_Event parity
against the oracle on the bench workload, on soups with quadrilaterals / exclusions / outside origins at several tree
shapes and on a 17-level tree; then kernel time at 1M rays.

    python tools/ab_oct.py persist:default pool:default pool:hare_amd/libhare_hip_x.so ...
"""
import os, subprocess, sys
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, zlib
sys.path.insert(0, %r)
import numpy as np, torch
import hare_amd as H
from oracle import pyoracle as po
from tests.helpers import soup, soup_rays
from tests.test_gpu_round2 import deep_scene
F = ("hit", "poly_id", "t", "x", "y", "z", "u", "v")
msgs = []
mesh = H.scenes.hall(); T = H.Topology(mesh.verts, mesh.nverts); g = H.Octree([T], 8, 16)
N = 1 << 20
rays = H.scenes.burst_rays(N, mesh.size)
ev, c = g.Shoot_batch(rays)
ref, rc = po.Octree([po.Topology(mesh.verts, mesh.nverts)], 8, 16).shoot(rays, nthreads=16)
bad = sum(int(np.count_nonzero(ev[f] != ref[f])) for f in F)
msgs.append("burst %%s (ctr %%d/%%d vs %%d)" %% ("OK" if bad == 0 else "DIFF %%d" %% bad, c["rays"], c["hits"], rc["hits"]))
v, nv, size = soup(); sr = soup_rays(20000, size)
rng = np.random.default_rng(1); e1 = rng.integers(-1, len(nv), len(sr)).astype(np.int32); e2 = rng.integers(-1, len(nv), len(sr)).astype(np.int32)
bad = 0
for depth, maxp in ((0, 4), (1, 1), (3, 2), (6, 8), (12, 64)):
    gs = H.Octree([H.Topology(v, nv)], depth, maxp); os_ = po.Octree([po.Topology(v, nv)], depth, maxp)
    for kw, okw in (({}, {}), ({"poly_origin1": e1, "poly_origin2": e2}, {"excl1": e1, "excl2": e2})):
        a, _ = gs.Shoot_batch(sr, **kw); b, _ = os_.shoot(sr, **okw)
        bad += sum(int(np.count_nonzero(a[f] != b[f])) for f in F)
msgs.append("soup %%s" %% ("OK" if bad == 0 else "DIFF %%d" %% bad))
dv, dnv, dr = deep_scene(17)
a, _ = H.Octree([H.Topology(dv, dnv)], 17, 1).Shoot_batch(dr); b, _ = po.Octree([po.Topology(dv, dnv)], 17, 1).shoot(dr)
bad = sum(int(np.count_nonzero(a[f] != b[f])) for f in F)
msgs.append("deep17 %%s" %% ("OK" if bad == 0 else "DIFF %%d" %% bad))
st = torch.cuda.current_stream().cuda_stream
for N in [int(x) for x in os.environ.get("RAYS", str(1 << 20)).split(",")]:
    rays = H.scenes.burst_rays(N, mesh.size)
    dr = torch.from_numpy(rays).cuda(); out = torch.empty(N * 56, dtype=torch.uint8, device="cuda")
    best = 1e9
    for rep in range(2):
        for _ in range(2): g.shoot_device(N, dr.data_ptr(), out.data_ptr(), stream=st)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1_ = torch.cuda.Event(enable_timing=True)
        e0.record()
        K = 5
        for _ in range(K): g.shoot_device(N, dr.data_ptr(), out.data_ptr(), stream=st)
        e1_.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1_) / K)
    msgs.append("n=%%d %%.3f ms %%.0f Mrays/s crc %%08x %%s" %% (N, best, N / best / 1e3, zlib.crc32(out.cpu().numpy().tobytes()), g.kernel_name(N)))
print(" | ".join(msgs))
''' % here
for spec in sys.argv[1:]:
    kern, lib = spec.split(":", 1)
    env = dict(os.environ, HARE_DEV="1", HARE_OCTREE_KERNEL=kern)
    if lib != "default": env["HARE_LIB"] = os.path.abspath(lib)
    try:
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=float(os.environ.get("AB_TIMEOUT", 200)))
    except subprocess.TimeoutExpired:
        print("%-36s TIMEOUT (hung kernel?) -- stopping" % spec, flush=True)
        sys.exit(3)      # never start another GPU run after a hang
    print("%-36s %s" % (spec, (r.stdout.strip().splitlines() or [r.stderr[-600:]])[-1]), flush=True)
    if r.returncode != 0:
        print(r.stderr[-1500:], flush=True)
