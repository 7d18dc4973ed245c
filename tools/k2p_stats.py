"""Developer: what a round of K2p's loop is made of (build with -DHARE_K2P_STATS: tools/build_variants.sh "pstats:-DHARE_K2P_STATS=1";
run with HARE_LIB=hare_amd/libhare_hip_pstats.so HARE_DEV=1 python tools/k2p_stats.py [rays]).  GPU box."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import hare_amd as H

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20          # K2d: "cull iterations" are dense passes, their "lanes" the items of a pass
m = H.scenes.hall()
g = H.Octree([H.Topology(m.verts, m.nverts)], 8, 16)
g.set_option("dev", 1)
g.set_option("octree_kernel", 4 if os.environ.get("KERNEL", "persist") == "dense" else 1)      # KERNEL=dense: K2d
g.set_option("octree_tail", 0)
rays = torch.from_numpy(H.scenes.burst_rays(n, m.size)).cuda()
out = torch.empty(n * 56, dtype=torch.uint8, device="cuda")
ctr = torch.zeros(8 + 64 + 4 * 4096 * 4, dtype=torch.int64, device="cuda")
sp = torch.cuda.current_stream().cuda_stream
g.shoot_device(n, rays.data_ptr(), out.data_ptr(), stream=sp)
ctr.zero_()
g.shoot_device(n, rays.data_ptr(), out.data_ptr(), d_counters=ctr.data_ptr(), stream=sp, flags=0x1000)
torch.cuda.synchronize()
c = ctr.cpu().numpy()
names = ["rounds", "alive", "P", "P_lanes", "C", "C_lanes", "E", "E_lanes"]
st = dict(zip(names, [float(x) for x in c[8:16]]))
r = float(c[0])
print("rays", int(r))
print("wave-rounds per ray %.3f; alive lanes per round %.1f" % (st["rounds"] / r, st["alive"] / st["rounds"]))
print("per round: P steps %.2f (lanes %.1f)  cull iterations %.2f (lanes %.1f)  exact %.2f (lanes %.1f)" %
      (st["P"] / st["rounds"], st["P_lanes"] / max(st["P"], 1), st["C"] / st["rounds"], st["C_lanes"] / max(st["C"], 1),
       st["E"] / st["rounds"], st["E_lanes"] / max(st["E"], 1)))
print("per ray: pops %.1f  cull pair-iterations %.1f  exact tests %.2f" % (st["P_lanes"] / r, st["C_lanes"] / r, st["E_lanes"] / r))
x = dict(zip(["int_steps", "int_lanes", "leaf_steps", "leaf_lanes", "exhausted", "dropped"], [float(v) for v in c[16:22]]))
print("pop steps with an interior visit %.2f per round at %.1f lanes; with a leaf visit %.2f at %.1f lanes" %
      (x["int_steps"] / st["rounds"], x["int_lanes"] / max(x["int_steps"], 1), x["leaf_steps"] / st["rounds"], x["leaf_lanes"] / max(x["leaf_steps"], 1)))
print("per ray: interior visits %.1f  leaf visits %.1f  exhausted-frame steps %.1f  pops dropped at the pop test %.1f" %
      (x["int_lanes"] / r, x["leaf_lanes"] / r, x["exhausted"] / r, x["dropped"] / r))
