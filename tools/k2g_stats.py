"""Developer: what an iteration of K2g's loop is made of (build with -DHARE_K2G_STATS, tools/build_variants.sh "stats:-DHARE_K2G_STATS=1";
run with HARE_LIB=hare_amd/libhare_hip_stats.so HARE_DEV=1 python tools/k2g_stats.py [rays]).  GPU box."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import hare_amd as H

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
m = H.scenes.hall()
g = H.Octree([H.Topology(m.verts, m.nverts)], 8, 16)
g.set_option("dev", 1)
g.set_option("octree_kernel", 3)
rays = torch.from_numpy(H.scenes.burst_rays(n, m.size)).cuda()
out = torch.empty(n * 56, dtype=torch.uint8, device="cuda")
ctr = torch.zeros(8 + 64, dtype=torch.int64, device="cuda")
sp = torch.cuda.current_stream().cuda_stream
g.shoot_device(n, rays.data_ptr(), out.data_ptr(), stream=sp)          # warm
ctr.zero_()
g.shoot_device(n, rays.data_ptr(), out.data_ptr(), d_counters=ctr.data_ptr(), stream=sp, flags=0x1000)
torch.cuda.synchronize()
c = ctr.cpu().numpy()
names = ["iter", "alive_groups", "pop", "pop_groups", "F", "F_groups", "L", "L_groups", "L_entries", "E", "E_lanes", "valid_hits"]
st = dict(zip(names, c[8:20]))
r = float(c[0])
print("rays", int(r), "hits", int(c[1]))
print("per ray: wave-iterations x8 = %.1f group-steps available; alive %.1f" % (st["iter"] * 8 / r, st["alive_groups"] / r))
print("per ray: pops %.2f  F %.2f  L chunks %.2f  entries %.1f  exact tests %.2f  valid hits %.2f" %
      (st["pop_groups"] / r, st["F_groups"] / r, st["L_groups"] / r, st["L_entries"] / r, st["E_lanes"] / r, st["valid_hits"] / r))
print("phase executions per wave-iteration: pop %.2f F %.2f L %.2f E %.2f" % (st["pop"] / st["iter"], st["F"] / st["iter"], st["L"] / st["iter"], st["E"] / st["iter"]))
print("groups active per execution: pop %.2f F %.2f L %.2f ; E lanes %.1f ; entries per L group-step %.2f" %
      (st["pop_groups"] / max(st["pop"], 1), st["F_groups"] / max(st["F"], 1), st["L_groups"] / max(st["L"], 1), st["E_lanes"] / max(st["E"], 1),
       st["L_entries"] / max(st["L_groups"], 1)))
print("wave-iterations per ray %.2f" % (st["iter"] / r))
