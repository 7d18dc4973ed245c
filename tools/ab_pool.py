"""Developer A/B for the voxel kernels: K1p (hare_voxel_persist_*) vs K1q (hare_voxel_pool_*, voxel_pool.hip), over
several builds of libhare_hip (HARE_LIB).  One subprocess per variant (a hung kernel stops the whole run, nothing is
retried).  Per variant: X_Event parity against the oracle on the bench workload AND on a soup with quadrilaterals,
outside origins, exclusions and origin write-back; then kernel time at 1M (and optionally more) rays.

    python tools/ab_pool.py persist:default pool:default pool:hare_amd/libhare_hip_w12.so ...
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
scene = os.environ.get("SCENE", "hall"); D = int(os.environ.get("DOMAIN", 64))
mesh = H.scenes.SCENES[scene](); T = H.Topology(mesh.verts, mesh.nverts); g = H.Voxel_Grid([T], D)
msgs = []
# parity 1: bench rays
N = 1 << 20
rays = H.scenes.burst_rays(N, mesh.size)
ev, c = g.Shoot_batch(rays)
ref, rc = po.VoxelGrid([po.Topology(mesh.verts, mesh.nverts)], domain=D).shoot(rays, nthreads=16)
bad = sum(int(np.count_nonzero(ev[f] != ref[f])) for f in ("hit", "poly_id", "t", "x", "y", "z", "u", "v"))
msgs.append("burst parity %%s (ctr %%d/%%d vs %%d)" %% ("OK" if bad == 0 else "DIFF %%d" %% bad, c["rays"], c["hits"], rc["hits"]))
# parity 2: soup with quads, outside origins, exclusions, write-back
v, nv, size = soup(); sr = soup_rays(20000, size)
rng = np.random.default_rng(1); e1 = rng.integers(-1, len(nv), len(sr)).astype(np.int32); e2 = rng.integers(-1, len(nv), len(sr)).astype(np.int32)
gs = H.Voxel_Grid([H.Topology(v, nv)], 12); os_ = po.VoxelGrid([po.Topology(v, nv)], domain=12)
r1 = sr.copy(); ev, _ = gs.Shoot_batch(r1, poly_origin1=e1, poly_origin2=e2, writeback_origin=True)
ref, _, moved = os_.shoot(sr, excl1=e1, excl2=e2, mutate=True)
bad = sum(int(np.count_nonzero(ev[f] != ref[f])) for f in ("hit", "poly_id", "t", "x", "y", "z", "u", "v")) + int(np.count_nonzero(r1 != moved))
msgs.append("soup parity %%s" %% ("OK" if bad == 0 else "DIFF %%d" %% bad))
st = torch.cuda.current_stream().cuda_stream
for N in [int(x) for x in os.environ.get("RAYS", str(1 << 20)).split(",")]:
    rays = H.scenes.burst_rays(N, mesh.size)
    dr = torch.from_numpy(rays).cuda(); out = torch.empty(N * 56, dtype=torch.uint8, device="cuda")
    best = 1e9
    for rep in range(3):
        for _ in range(3): g.shoot_device(N, dr.data_ptr(), out.data_ptr(), stream=st)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1_ = torch.cuda.Event(enable_timing=True)
        e0.record()
        K = max(3, min(30, (30 << 20) // N))
        for _ in range(K): g.shoot_device(N, dr.data_ptr(), out.data_ptr(), stream=st)
        e1_.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1_) / K)
    msgs.append("n=%%d %%.4f ms %%.0f Mrays/s crc %%08x" %% (N, best, N / best / 1e3, zlib.crc32(out.cpu().numpy().tobytes())))
print(" | ".join(msgs))
''' % here
for spec in sys.argv[1:]:
    kern, lib = spec.split(":", 1)
    env = dict(os.environ, HARE_DEV="1", HARE_VOXEL_KERNEL=kern)
    if lib != "default": env["HARE_LIB"] = os.path.abspath(lib)
    try:
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=float(os.environ.get("AB_TIMEOUT", 150)))
    except subprocess.TimeoutExpired:
        print("%-44s TIMEOUT (hung kernel?) -- stopping" % spec, flush=True)
        sys.exit(3)      # never start another GPU run after a hang
    print("%-44s %s" % (spec, (r.stdout.strip().splitlines() or [r.stderr[-600:]])[-1]), flush=True)
    if r.returncode != 0:
        print(r.stderr[-1500:], flush=True)
