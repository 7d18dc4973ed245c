"""Developer A/B: time the bench workload with several builds of libhare_hip (HARE_LIB), one subprocess each."""
import os, subprocess, sys
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch
import hare_amd as H
N = int(os.environ.get("RAYS", 1 << 20)); D = int(os.environ.get("DOMAIN", 64))
mesh = H.scenes.hall(); g = H.Voxel_Grid([H.Topology(mesh.verts, mesh.nverts)], D)
rays = H.scenes.burst_rays(N, mesh.size)
dr = torch.from_numpy(rays).cuda(); out = torch.empty(N * 56, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
best = 1e9
for rep in range(3):
    for _ in range(3): g.shoot_device(N, dr.data_ptr(), out.data_ptr(), stream=st)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): g.shoot_device(N, dr.data_ptr(), out.data_ptr(), stream=st)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 30)
import zlib
print("%%.4f ms  %%.0f Mrays/s  crc %%08x" %% (best, N / best / 1e3, zlib.crc32(out.cpu().numpy().tobytes())))
''' % here
for lib in sys.argv[1:]:
    env = dict(os.environ)
    if lib != "default": env["HARE_LIB"] = os.path.abspath(lib)
    try:
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=float(os.environ.get("AB_TIMEOUT", 90)))
    except subprocess.TimeoutExpired:
        print("%-40s TIMEOUT (hung kernel?) -- stopping" % os.path.basename(lib), flush=True)
        sys.exit(3)      # never start another GPU run after a hang
    print("%-40s %s" % (os.path.basename(lib), (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1]), flush=True)
