"""Developer A/B for the octree kernel: like ab_libs.py but times Octree(8, 16) on the bench workload."""
import os, subprocess, sys
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, zlib
sys.path.insert(0, %r)
import numpy as np, torch
import hare_amd as H
N = int(os.environ.get("RAYS", 1 << 20))
mesh = H.scenes.hall(); g = H.Octree([H.Topology(mesh.verts, mesh.nverts)], 8, 16)
rays = H.scenes.burst_rays(N, mesh.size)
dr = torch.from_numpy(rays).cuda(); out = torch.empty(N * 56, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
best = 1e9
for rep in range(2):
    g.shoot_device(N, dr.data_ptr(), out.data_ptr(), stream=st); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8): g.shoot_device(N, dr.data_ptr(), out.data_ptr(), stream=st)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 8)
print("%%.3f ms  %%.1f Mrays/s  crc %%08x" %% (best, N / best / 1e3, zlib.crc32(out.cpu().numpy().tobytes())))
''' % here
for lib in sys.argv[1:]:
    env = dict(os.environ)
    if lib != "default": env["HARE_LIB"] = os.path.abspath(lib)
    try:
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=float(os.environ.get("AB_TIMEOUT", 120)))
    except subprocess.TimeoutExpired:
        print("%-40s TIMEOUT (hung kernel?) -- stopping" % os.path.basename(lib), flush=True)
        sys.exit(3)
    print("%-40s %s" % (os.path.basename(lib), (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1]), flush=True)
