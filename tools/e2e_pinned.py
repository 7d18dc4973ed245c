"""Developer experiment: hare_shoot_batch from host buffers (H2D + kernel + D2H), pageable vs page-locked (hipHostRegister)
caller buffers, 1M and 4M rays, hall D=64.  The C-ABI call is the same; only what kind of memory the caller hands over differs."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hare_amd as H

mesh = H.scenes.hall(); g = H.Voxel_Grid([H.Topology(mesh.verts, mesh.nverts)], 64)
rt = torch.cuda.cudart()
for n in (1 << 20, 1 << 22):
    rays = H.scenes.burst_rays(n, mesh.size)
    ev = np.zeros(n, H.capi.XEVENT_DTYPE)
    ctr = H.capi.Counters()

    def run(K=5):
        best = 1e9
        for _ in range(K):
            t0 = time.perf_counter()
            H.capi.check(H.capi.lib.hare_shoot_batch(g._h, g._kind, 0, n, rays.ctypes.data, None, None, 0, ev.ctypes.data, C.byref(ctr)))
            best = min(best, time.perf_counter() - t0)
        return best
    run(2)
    t_page = run()
    crc_page = int(np.bitwise_xor.reduce(ev.view(np.uint64).reshape(-1)))
    t0 = time.perf_counter()
    assert int(rt.cudaHostRegister(rays.ctypes.data, rays.nbytes, 0)) == 0
    assert int(rt.cudaHostRegister(ev.ctypes.data, ev.nbytes, 0)) == 0
    t_reg = time.perf_counter() - t0
    ev[:] = 0
    run(2)
    t_pin = run()
    crc_pin = int(np.bitwise_xor.reduce(ev.view(np.uint64).reshape(-1)))
    rt.cudaHostUnregister(rays.ctypes.data); rt.cudaHostUnregister(ev.ctypes.data)
    print("n=%d: pageable %.3f ms (%.0f Mrays/s) | registered %.3f ms (%.0f Mrays/s), registering both buffers took %.1f ms | events equal: %s | hits %d"
          % (n, t_page * 1e3, n / t_page / 1e6, t_pin * 1e3, n / t_pin / 1e6, t_reg * 1e3, crc_page == crc_pin, ctr.hits))
    for chunks in (1, 2, 3):
        g.set_option("batch_chunks", chunks)
        assert int(rt.cudaHostRegister(rays.ctypes.data, rays.nbytes, 0)) == 0 and int(rt.cudaHostRegister(ev.ctypes.data, ev.nbytes, 0)) == 0
        run(1); t = run()
        rt.cudaHostUnregister(rays.ctypes.data); rt.cudaHostUnregister(ev.ctypes.data)
        print("   registered, %d chunk(s): %.3f ms (%.0f Mrays/s)" % (chunks, t * 1e3, n / t / 1e6))
    g.set_option("batch_chunks", 0)
