#!/usr/bin/env python3
"""bench.py -- Mrays/s of the Hare ray-cast hot path on MI355X (BASELINE.json metric).

A step = one pass of the hot path over one batch: `rays_per_gpu` primary rays (spherical-Fibonacci
burst) cast into the 100k-triangle hall through Voxel_Grid's 3D-DDA kernel (BASELINE.json
configs[1]), inputs and outputs resident in HBM.  With N GPUs the burst is sharded contiguously
(rank r casts rays [r*n, (r+1)*n) of an N*n-ray burst: weak scaling, scene replicated, no data-path
collective); the only exchange is the RCCL all-reduce of the hit counter after each step.

    python bench.py --gpus N --steps K --warmup W          # N > 1: starts N ranks itself (one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Other configs: --kind octree (config 3), --scene cathedral --domain 128 --rays 2097152 (config 4 shard),
--scene cathedral --domain 128 --bounces 8 (config 5: device-resident specular bounce loop, value = casts/s),
--kind kdtree --scene shoebox (KDTree.Shoot, a brute-force query in the reference: one measured line, profiles/).
The default run (the driver's command, at ANY N) measures the headline workload first and then, in the same process(es), the rest
of BASELINE's table at fewer steps, attached to the ONE JSON line as "configs": {...}, each entry with its own value, roofline,
parity flag and (N = 1) cpu_baseline:
    c4, c5                   configs 4 and 5 as stated: ONE 16 777 216-ray burst / ONE 8 388 608-ray burst x 8 specular bounces into
                             the 1M-triangle cathedral, contiguous shards over the N ranks ("scaling": "strong"), hit counters
                             all-reduced (RCCL), max-over-ranks timing, every rank's X_Events checked against the oracle
    c3, c4_shard, c5_shard   N = 1 only: config 3, and one GPU's share of configs 4 / 5 at N = 8 (2M rays; 1M rays x 8)
    c2_quads                 N = 1 only: the headline workload on `hall_quads` -- the hall with its flat lattices un-split, 39k planar
                             quadrilaterals + 22k triangles: the quadrilateral build of the voxel kernel (hare_voxel_pool_quad)
(--no-extra-configs skips them).  --force-dist runs the torch.distributed code path (init, per-step async all-reduce of
the counters, all-gather of the timings) even at N = 1, so that the RCCL branch can be exercised on a one-GPU box.

Prints ONE JSON line on rank 0.  `roofline.achieved` = algorithmic bytes per launch (SURVEY.md 8(d):
104 + 8*C + 4*L + 96*T per ray for the grid, 64*C for the octree, + 28 B per bounce; C/L/T counted exactly
by the oracle on the same rays) / average duration of the shoot kernel measured here with HIP events on the
launch stream.  `cpu_baseline` = the oracle (C restatement of the reference's Shoot, kind "port") timed on
this box's host cores.  The oracle is the checker and the CPU baseline only; the timed path never touches it.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
KERNEL_SOURCES = ("hare_amd/csrc/kernels.hip", "hare_amd/csrc/voxel_pool.hip", "hare_amd/csrc/voxel_walk.h", "hare_amd/csrc/voxel_coop.hip", "hare_amd/csrc/octree_pool.hip",
                  "hare_amd/csrc/octree_coop.hip", "hare_amd/csrc/octree_group.hip", "hare_amd/csrc/kdtree_dense.hip", "hare_amd/csrc/order_kernels.hip", "hare_amd/csrc/hare_math.h",
                  "hare_amd/csrc/hare_trace.h", "hare_amd/csrc/hare_device.h")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rays", type=int, default=1 << 20, help="rays per GPU per step")
    ap.add_argument("--domain", type=int, default=64, help="Voxel_Grid Domain")
    ap.add_argument("--scene", default="hall", choices=["hall", "cathedral", "shoebox", "hall_quads"])
    ap.add_argument("--kind", default="voxel", choices=["voxel", "octree", "kdtree"])
    ap.add_argument("--bounces", type=int, default=1,
                    help="casts per step: >1 = device-resident specular bounce loop (BASELINE config 5)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle pass (no roofline / cpu_baseline / parity)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the host-buffer (PCIe-inclusive) leg: profiling runs want only full-size launches")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo: rehearsal with ranks sharing one GPU)")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="N = 1, default workload: do not append configs 3, 4 (shard) and 5 to the line")
    ap.add_argument("--extra-configs", action="store_true", help="append configs 3, 4 (shard), 5 whatever the headline workload is")
    ap.add_argument("--extra-rays", type=int, default=0, help="cap the rays of the appended configs (tests; 0 = their real sizes)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed and run its code path even with one rank (tests the RCCL branch on one GPU)")
    ap.add_argument("--inproc-scenes", type=int, default=-1,
                    help="the single-process multi-GPU leg (`inproc_sharded`): ONE hare_shoot_batch_sharded and ONE hare_bounce_batch_sharded call "
                         "over this many scenes, scene k on device k %% device_count.  -1 (default): one scene per visible device when N > 1, off at "
                         "N = 1; 0: off; K > 0: K scenes (a one-GPU box runs it with two scenes on its one device)")
    ap.add_argument("--bounce-api", default="device", choices=["device", "batch"],
                    help="--bounces > 1: 'device' = hare_shoot_device + hare_reflect_device on resident buffers (the timed loop); "
                         "'batch' additionally times hare_bounce_batch from host buffers (PCIe-inclusive, reported beside it)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ N > 1 launcher
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(world: int, argv, script: str | None = None) -> int:
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: start N ranks (one process per GPU) as
    children, forward rank 0's JSON line, fail if any rank fails.  This process never touches the GPU.
    (`script`: the rank program, this file by default; the launcher test passes a stand-in.)"""
    port = _free_port()
    script = os.path.abspath(script or __file__)
    procs = []
    for r in range(world):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    lines = []

    def pump():
        for ln in procs[0].stdout:
            lines.append(ln)

    t = threading.Thread(target=pump, daemon=True)
    t.start()
    failed = None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = bad[0]
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.05)
    if failed is not None:
        for p in procs:            # exact PIDs we started, never a pattern
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
        sys.stderr.write(f"bench.py: rank {failed[0]} exited with code {failed[1]}\n")
        return 1
    t.join(timeout=10)
    # stdout carries the ONE JSON line; anything else rank 0 (or a library under it) printed goes to stderr
    for ln in lines:
        (sys.stdout if ln.lstrip().startswith("{") else sys.stderr).write(ln)
    sys.stdout.flush()
    if not any(ln.lstrip().startswith("{") for ln in lines):
        sys.stderr.write("bench.py: rank 0 printed no JSON line\n")
        return 1
    return 0


# ------------------------------------------------------------------------------------------------ helpers
def algorithmic_bytes(ctr: dict, kind: str, bounce_links: int = 0) -> int:
    """SURVEY.md 8(d) / BASELINE.md 4: B = sum over casts of 104 + cw*C + 4*L + 96*T (cw = 8 B per grid cell,
    64 B per octree / kd node) + 28 B per bounce (normal read + exclusion write)."""
    cw = 8 if kind == "voxel" else 64
    return 104 * ctr["rays"] + cw * ctr["cells"] + 4 * ctr["entries"] + 96 * ctr["tests"] + 28 * bounce_links


def kernel_source_sha() -> str:
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def load_traffic(workload_key: str):
    """The committed rocprofv3 PMC passes for this workload (profiles/traffic.json): HBM bytes per launch and the issue-side
    counters.  An entry is only reported when it was measured on THIS kernel source (its `kernel_sha16`), else None:
    counters of an older kernel say nothing about the one that just ran."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(p) as f:
            t = json.load(f)
        e = t.get(workload_key)
        if e is None or e.get("kernel_sha16") != kernel_source_sha():
            return None
        return e
    except Exception:
        return None


N_SIMDS = 1024            # 256 CUs x 4 SIMDs
# What a wave64 VALU instruction costs a SIMD, by class: ns per instruction per SIMD with four waves resident, MEASURED on the MI355X
# box with tools/valu_rate.hip (independent instructions; profiles/r05_experiments/valu_rate.log).  FP64 runs at half the FP32 / integer
# rate (16 lanes per clock and SIMD), v_rcp_f64 at a sixth.  `roofline.issue` prices the PMC instruction counts with these.
VALU_NS = {"add_f64": 1.98, "mul_f64": 2.32, "fma_f64": 2.61, "trans_f64": 6.95, "other": 1.30}


def kd_params(scene: str):
    """KDTree(maxDepth, maxPolygonsPerNode) per scene: the shoebox as round 4 measured it; the 100k-polygon scenes 16 / 8."""
    return (12, 16) if scene == "shoebox" else (16, 8)


class Env:
    """What every workload of one bench process shares: the rank, the (optional) process group, the device."""

    def __init__(self, args, rank, local_rank, world, dist, device):
        self.args, self.rank, self.local_rank, self.world, self.dist, self.device = args, rank, local_rank, world, dist, device
        self.backend = args.backend
        self.meshes = {}       # scene name -> (mesh, H.Topology)
        self.parts = {}        # (scene, kind, domain) -> (partition, description, build seconds)
        self.oracles = {}      # (scene, kind, domain) -> (oracle topology, oracle partition, reference name)

    def mesh(self, H, scene):
        if scene not in self.meshes:
            m = H.scenes.SCENES[scene]()
            self.meshes[scene] = (m, H.Topology(m.verts, m.nverts))
        return self.meshes[scene]

    def partition(self, H, scene, kind, domain):
        key = (scene, kind, domain if kind == "voxel" else 0)
        if key not in self.parts:
            _, topo = self.mesh(H, scene)
            t0 = time.time()
            if kind == "voxel":
                part, kdesc = H.Voxel_Grid([topo], domain, device=self.device), f"Voxel_Grid Domain={domain}"
            elif kind == "octree":
                part, kdesc = H.Octree([topo], 8, 16, device=self.device), "Octree maxDepth=8 maxPolys=16"
            else:
                kd_d, kd_p = kd_params(scene)
                part, kdesc = H.KDTree([topo], kd_d, kd_p, device=self.device), f"KDTree maxDepth={kd_d} maxPolys={kd_p}"
            self.parts[key] = (part, kdesc, time.time() - t0)
        return self.parts[key]

    def oracle(self, H, scene, kind, domain):
        """The checker (oracle/): only ever called after the timed region, on rank 0."""
        from oracle import pyoracle as po
        key = (scene, kind, domain if kind == "voxel" else 0)
        if key not in self.oracles:
            m, _ = self.mesh(H, scene)
            ot = po.Topology(m.verts, m.nverts)
            if kind == "voxel":
                og, ref_name = po.VoxelGrid([ot], domain=domain), "Voxel_Grid.Shoot"
            elif kind == "octree":
                og, ref_name = po.Octree([ot], 8, 16), "Octree.Shoot"
            else:
                og, ref_name = po.KDTree([ot], *kd_params(scene)), "KDTree.Shoot"
            self.oracles[key] = (ot, og, ref_name)
        return self.oracles[key]

    def drop(self, scene=None):
        """Free partitions (device memory) of `scene`, or all."""
        for key in [k for k in self.parts if scene is None or k[0] == scene]:
            self.parts.pop(key)[0].close()
        for key in [k for k in self.oracles if scene is None or k[0] == scene]:
            self.oracles.pop(key)


def measure(w, env):
    """One workload (scene, partition, rays per GPU, bounces) measured per the bench contract: `warmup` untimed steps, then
    exactly `steps` steps between barrier + synchronize fences, MAX over ranks.  Returns the JSON line (rank 0) or None."""
    import numpy as np
    import torch

    import hare_amd as H
    from hare_amd.sharding import shard_range

    rank, world, dist, backend = env.rank, env.world, env.dist, env.backend
    scene, kind, domain = w["scene"], w["kind"], w["domain"]
    steps, warmup = w["steps"], w["warmup"]
    mesh, _ = env.mesh(H, scene)
    part, kdesc, build_s = env.partition(H, scene, kind, domain)

    B = w["bounces"]
    strong = bool(w.get("rays_total"))
    if strong:        # BASELINE configs 4 / 5: ONE batch of rays_total rays, cut into contiguous shards (strong scaling)
        n_total = int(w["rays_total"])
        lo, hi = shard_range(n_total, rank, world)
        n = hi - lo
    else:             # the headline: every GPU casts its own n rays of an N*n-ray burst (weak scaling)
        n = w["rays"]
        n_total = n * world
        lo, hi = shard_range(n_total, rank, world)
    kernel_name = part.kernel_name(n)
    rays_h = H.scenes.burst_rays(n_total, mesh.size, start=lo, count=hi - lo)
    d_rays = torch.from_numpy(rays_h).cuda()
    d_out = torch.empty(n * 56, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream()
    sp = stream.cuda_stream
    # A caller streams NEW rays in and events out every batch.  Re-casting one pair of buffers would leave them warm in the
    # 256 MiB Infinity Cache from step to step; the timed steps therefore rotate through enough copies (same rays) that a
    # buffer's lines have been evicted by the time it comes round again: rays are read from HBM, events written to HBM.
    per_set = n * (48 + 56)
    n_sets = 1 if B > 1 else max(2, min(8, -(-640 * 1024 * 1024 // per_set)))
    ray_sets = [d_rays] + [d_rays.clone() for _ in range(n_sets - 1)]
    out_sets = [d_out] + [torch.empty_like(d_out) for _ in range(n_sets - 1)]

    d_rays0 = d_rays.clone() if B > 1 else None
    d_excl = torch.zeros(2 * n, dtype=torch.int32, device="cuda") if B > 1 else None       # hare_bounce_device's work array
    # the per-batch hit-count reduce runs on RCCL's stream, overlapped with the NEXT batch's kernel:
    # two counter blocks alternate, a block is reused only after its all-reduce has been waited for
    ctrs = [torch.zeros(8, dtype=torch.int64, device="cuda"), torch.zeros(8, dtype=torch.int64, device="cuda")]
    # a block starts every step as {0, ..., 0, 1}: the kernels accumulate rays / hits into words 0 / 1 and never touch word 7
    # (hare_counters.reserved[2]), so after the reduce word 7 is the number of ranks whose block really was summed
    ctr_init = torch.tensor([0, 0, 0, 0, 0, 0, 0, 1], dtype=torch.int64, device="cuda")
    pending = [None, None]
    reduced = [None, None]
    state = {"k": 0, "set": 0}

    def cast_pass(c_ptr, events=None):
        """One pass of the hot path over this rank's batch: 1 cast, or B casts with a specular bounce between them."""
        if B > 1:     # config 5: shoot -> reflect -> shoot with poly_origin1 = the polygon just hit, B casts, device-resident:
            # hare_bounce_device -- ONE launch in which every ray runs through its casts (Voxel_Grid), or a launch per cast (trees)
            d_rays.copy_(d_rays0)        # the call overwrites its rays (work array): every step starts from the burst
            if events is not None:
                events[0].record(stream)
            part.bounce_device(n, d_rays.data_ptr(), B, d_excl.data_ptr(), d_events_last=d_out.data_ptr(), d_counters=c_ptr, stream=sp)
            if events is not None:
                events[1].record(stream)
        else:
            i = state["set"] = (state["set"] + 1) % n_sets
            if events is not None:
                events[0].record(stream)
            part.shoot_device(n, ray_sets[i].data_ptr(), out_sets[i].data_ptr(), d_counters=c_ptr, stream=sp)
            if events is not None:
                events[1].record(stream)

    def step():
        k = state["k"]
        state["k"] = k + 1
        c = ctrs[k & 1]
        if pending[k & 1] is not None:
            pending[k & 1].wait()
            pending[k & 1] = None
        if dist is not None:
            c.copy_(ctr_init)          # a block is all-reduced per step: it starts every step from {0, ..., 0, 1}
        cast_pass(c.data_ptr())        # (one rank, no reduce: the library ACCUMULATES into the caller's counters; they run on over the timed steps)
        if dist is not None:
            if backend == "nccl":
                pending[k & 1] = dist.all_reduce(c, async_op=True)   # RCCL: the final hit-count reduce (64 B)
                reduced[k & 1] = c
            else:                                                    # gloo rehearsal: through the host
                h = c.cpu()
                dist.all_reduce(h)
                reduced[k & 1] = h
        else:
            reduced[k & 1] = c

    def drain():
        for i in (0, 1):
            if pending[i] is not None:
                pending[i].wait()
                pending[i] = None

    def fence():
        drain()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up, not a step: the first launch loads the code object and sizes the scene's scratch
    part.shoot_device(n, d_rays.data_ptr(), d_out.data_ptr(), stream=sp)
    for _ in range(warmup):
        step()
    fence()
    if dist is None:
        for c in ctrs:                 # outside the timed region: the timed steps accumulate from zero
            c.zero_()
        state["k"] = 0
        torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t_start = time.perf_counter()
    ev0.record(stream)
    for _ in range(steps):
        step()
    ev1.record(stream)
    fence()
    wall = time.perf_counter() - t_start
    dev_ms = ev0.elapsed_time(ev1)
    walls = [wall]
    if dist is not None:
        tw = torch.tensor([wall, dev_ms], dtype=torch.float64)
        if backend == "nccl":
            tw = tw.cuda()
        gathered = [torch.zeros_like(tw) for _ in range(world)]
        dist.all_gather(gathered, tw)
        walls = [float(g[0]) for g in gathered]
        dev_ms = max(float(g[1]) for g in gathered)
    wall = max(walls)
    if dist is None:
        # the two blocks hold the sums over their steps; every step casts the same rays
        tot = (ctrs[0] + ctrs[1]).cpu()
        assert int(tot[0]) % steps == 0 and int(tot[1]) % steps == 0, "counters are not a whole number of steps"
        hits_total, rays_total, ranks_seen = int(tot[1]) // steps, int(tot[0]) // steps, 1
    else:
        last = reduced[(state["k"] - 1) & 1]
        hits_total = int(last[1])
        rays_total = int(last[0])
        ranks_seen = int(last[7])

    # shoot-kernel duration: HIP events around each shoot launch on the launch stream (the stream the kernel runs on)
    nrep = max(1, min(steps, 30))
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(2 * B)] for _ in range(nrep)]
    cast_pass(0)
    torch.cuda.synchronize()
    for r in range(nrep):
        cast_pass(0, evs[r])
    torch.cuda.synchronize()
    loop_ms = sum(evs[r][0].elapsed_time(evs[r][1]) for r in range(nrep)) / nrep      # one shoot launch, or the whole bounce loop
    kern_ms_pairs = loop_ms / B               # average per cast; an event pair around every call
    # ONE estimator of the kernel's duration, stated in the line (ADVICE, round 4: not the smaller of two):
    #   one rank, one cast per step: a timed step IS one shoot launch and nothing else (the counters run on, no reset kernel), so the two
    #   HIP events around the K timed launches give the average launch duration over the timed region itself -- what the contract asks
    #   for, and what rocprofv3's average for the kernel agrees with (profiles/);
    #   anything else (a reduce per step, a bounce loop): the event pair around every call.
    if B == 1 and dist is None:
        kern_ms, kern_src = dev_ms / steps, "timed region / steps (HIP events)"
    else:
        kern_ms, kern_src = kern_ms_pairs, "event pair per call (HIP events)"
    per_cast_ms = [kern_ms] * B
    events_dev = out_sets[state["set"]].cpu().numpy().tobytes() if B == 1 else None   # the bench buffers themselves, for the parity check

    # Beside the contract's figure (steps on ONE stream, so that a launch's duration is that of an undisturbed launch): what a caller that
    # streams batches over TWO HIP streams gets -- one launch's drain runs under the next one's ramp (DESIGN.md 6 "Two streams").  Same
    # launches, same buffers in rotation, same events; reported as `two_streams`, never as `value`.
    two_streams = None
    if B == 1 and dist is None and w.get("two_streams", False) and n_sets >= 2 and w["cpu_baseline"]:      # (not in the profiling runs, --no-cpu-baseline:
        # launches that overlap would enter rocprofv3's per-kernel average, which is to agree with kernel_ms)
        ss = [torch.cuda.Stream(), torch.cuda.Stream()]
        k2 = max(4, steps)
        ne = n_sets - (n_sets % 2)                 # an even rotation: a buffer set always meets the same stream
        t0e = torch.cuda.Event(enable_timing=True)
        t1e = torch.cuda.Event(enable_timing=True)
        for rep_ in range(2):                      # the first repetition warms the streams up
            torch.cuda.synchronize()
            t0e.record(stream)
            for s_ in ss:
                s_.wait_event(t0e)
            for k in range(k2):
                part.shoot_device(n, ray_sets[k % ne].data_ptr(), out_sets[k % ne].data_ptr(), stream=ss[k % 2].cuda_stream)
            for s_ in ss:
                stream.wait_stream(s_)
            t1e.record(stream)
            torch.cuda.synchronize()
        t2 = t0e.elapsed_time(t1e)
        two_streams = {"value": round(n * k2 / t2 / 1e3, 2), "unit": "Mrays/s", "ms_per_step": round(t2 / k2, 4), "steps": k2,
                       "note": "same launches alternating over two HIP streams; not the contract's value"}

    # measured device-copy bandwidth (what "HBM peak" means in practice on this box) and the host-buffer (PCIe-inclusive) rate
    copy_gbs = None
    e2e = None
    e2e_slim = None
    slim_ok = None
    bounce_batch = None
    if rank == 0:
        if w.get("copy_bw", True):
            a = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
            b_ = torch.empty_like(a)
            b_.copy_(a)
            torch.cuda.synchronize()
            c0 = torch.cuda.Event(enable_timing=True)
            c1 = torch.cuda.Event(enable_timing=True)
            c0.record(stream)
            for _ in range(10):
                b_.copy_(a)
            c1.record(stream)
            torch.cuda.synchronize()
            copy_gbs = 2 * a.numel() * 10 / (c0.elapsed_time(c1) * 1e-3) / 1e9
            del a, b_
        if B == 1 and w["e2e"]:
            # host buffers in, host buffers out (H2D + kernel + D2H inside one hare_shoot_batch): the PCIe-inclusive rate
            import ctypes as C
            ev_h = np.zeros(n, H.capi.XEVENT_DTYPE)
            ctr_h = H.capi.Counters()

            def host_call():
                H.capi.check(H.capi.lib.hare_shoot_batch(part._h, part._kind, 0, n, rays_h.ctypes.data, None, None, 0, ev_h.ctypes.data,
                                                         C.addressof(ctr_h)))

            host_call()                         # sizes the scene's staging buffers: not part of the measurement
            best = None
            for _ in range(5):
                t1 = time.perf_counter()
                host_call()
                dt = time.perf_counter() - t1
                best = dt if best is None else min(best, dt)
            e2e = n / best / 1e6
            # the same call with slim result records (HARE_SHOOT_SLIM_EVENTS): 16 B (trees: 32 B) per ray come back instead of 56
            slim_dt = H.capi.SLIM_DTYPE if kind == "voxel" else H.capi.SLIM_UV_DTYPE
            sl_h = np.zeros(n, slim_dt)

            def host_call_slim():
                H.capi.check(H.capi.lib.hare_shoot_batch(part._h, part._kind, 0, n, rays_h.ctypes.data, None, None, H.capi.SHOOT_SLIM_EVENTS,
                                                         sl_h.ctypes.data, C.addressof(ctr_h)))

            host_call_slim()
            best = None
            for _ in range(5):
                t1 = time.perf_counter()
                host_call_slim()
                dt = time.perf_counter() - t1
                best = dt if best is None else min(best, dt)
            e2e_slim = n / best / 1e6
            slim_ok = bool(part.expand_events(rays_h, sl_h).tobytes() == ev_h.tobytes())     # rebuilt X_Events == the full call's, byte for byte
        if B > 1 and w.get("bounce_api") == "batch" and hasattr(part, "Bounce_batch"):
            # the same loop through ONE C-ABI call from host buffers (hare_bounce_batch): H2D once, B casts and B-1 reflections on
            # the device, the final events D2H -- what a C# / Pachyderm caller without device pointers gets
            part.Bounce_batch(rays_h, B)        # sizes the staging buffers
            best = None
            for _ in range(2):
                t1 = time.perf_counter()
                bb_ev, bb_ctr = part.Bounce_batch(rays_h, B)
                dt = time.perf_counter() - t1
                best = dt if best is None else min(best, dt)
            bounce_batch = {"mcasts_s": round(bb_ctr["rays"] / best / 1e6, 1), "ms": round(best * 1e3, 3), "casts": bb_ctr["rays"],
                            "events": bb_ev}

    # ---- the checker (oracle/), after the timed region.  Rank 0 passes its WHOLE shard through the oracle: exact C/L/T for the
    # roofline, the reference result for the parity check and (N = 1) the CPU baseline.  Every other rank checks a stated sample
    # of ITS shard the same way, so that an N-GPU line says that every device returned the reference's X_Events.
    roofline = None
    cpu = None
    parity = None
    parity_rays = None
    parity_ranks = None
    fields = ("hit", "poly_id", "t", "x", "y", "z", "u", "v")
    if w["cpu_baseline"]:
        from oracle import pyoracle as po
        affinity = len(os.sched_getaffinity(0))
        cores = int(os.environ.get("HARE_CPU_THREADS", "0")) or min(affinity, 256)    # the oracle takes up to 256 threads
        ot, og, ref_name = env.oracle(H, scene, kind, domain)

        def oracle_pass(rays, nthreads):
            """The same pass on the CPU: returns (events of the last cast, summed counters, casts with a live ray)."""
            tot = {"rays": 0, "hits": 0, "cells": 0, "entries": 0, "tests": 0}
            r = rays
            excl = None
            ev = None
            links = 0
            for b in range(B):
                if excl is None:
                    ev, c = og.shoot(r, nthreads=nthreads)
                else:
                    live = excl >= 0
                    ev = np.zeros(len(r), po.XEVENT_DTYPE)
                    ev["poly_id"] = -1
                    if live.any():
                        ev_l, c = og.shoot(r[live], excl1=excl[live], nthreads=nthreads)
                        ev[live] = ev_l
                    else:
                        c = {k: 0 for k in tot}
                for k in tot:
                    tot[k] += c[k]
                if b + 1 < B:
                    r = po.reflect_batch(ot, r, ev)
                    excl = np.where(ev["hit"] != 0, ev["poly_id"], -2).astype(np.int32)
                    links += int((ev["hit"] != 0).sum())
            return ev, tot, links

        def device_events(m):
            """The first m X_Events of the bench buffers themselves (B > 1: of the last cast)."""
            if events_dev is not None:
                return np.frombuffer(events_dev, dtype=H.capi.XEVENT_DTYPE)[:m]
            return np.frombuffer(d_out[:m * 56].cpu().numpy().tobytes(), dtype=H.capi.XEVENT_DTYPE)

    if w["cpu_baseline"] and rank != 0:
        m = min(n, PARITY_SAMPLE_RAYS)
        ref_s, _, _ = oracle_pass(rays_h[:m], min(cores, 32))
        got_s = device_events(m)
        parity = bool(all(np.array_equal(got_s[f], ref_s[f]) for f in fields))

    if w["cpu_baseline"] and rank == 0:
        # How many threads serve this workload best is measured, not assumed: every worker of the port keeps a mailbox of P entries
        # (as the reference's pool does per ThreadID), so on the 1M-triangle scene 256 threads are slower than 64.  A short sample at
        # a few thread counts picks the count the timed passes use; `cores` reports it beside the box's core count.
        # KDTree.Shoot tests EVERY polygon for every ray in the reference (F4): on a scene of more than a few thousand polygons the
        # oracle gets every 2^k-th ray of the batch (at most 65 536, all polar bands of the burst) -- parity on those, the reference's
        # C / L / T scaled up to the batch (they are the same for every ray: the whole tree), the CPU baseline timed on them
        samp = None
        if kind == "kdtree" and mesh.P > 5000 and n > 65536 and B == 1:
            samp = np.arange(0, n, n // 65536)
        rays_o = rays_h if samp is None else np.ascontiguousarray(rays_h[samp])
        if world == 1 and not os.environ.get("HARE_CPU_THREADS") and cores > 32 and samp is None:
            ns = min(n, 262144)
            trial = {}
            for nt in sorted({cores, max(32, cores // 2), max(32, cores // 4), 32}, reverse=True):
                c0 = time.perf_counter()
                oracle_pass(rays_h[:ns], nt)
                trial[nt] = time.perf_counter() - c0
            cores = min(trial, key=trial.get)
        # full pass: exact counters for the roofline + the reference result for the parity check; timed as the CPU baseline
        best = None
        budget = time.time() + w.get("cpu_budget_s", 20.0)
        reps = 0
        max_reps = w.get("cpu_reps", 5) if world == 1 else 1   # the CPU baseline is reported at N = 1 only
        ref = ctr = links = None
        while reps < max_reps and (reps < 1 or time.time() < budget):
            c0 = time.perf_counter()
            ref, ctr, links = oracle_pass(rays_o, cores)
            dt = time.perf_counter() - c0
            best = dt if best is None else min(best, dt)
            reps += 1
        cpu_casts = ctr["rays"]
        if samp is not None:
            ctr = {k: v * n // len(samp) for k, v in ctr.items()}
        casts = ctr["rays"]
        unit = "Mrays/s" if B == 1 else "Mcasts/s"
        if world == 1:
            n1 = min(n, (100000 if kind == "voxel" else (20000 if samp is None else 256)) // B)
            c0 = time.perf_counter()
            _, c1ctr, _ = oracle_pass(rays_h[:n1], 1)
            dt1 = time.perf_counter() - c0
            cpu = {"value": round(cpu_casts / best / 1e6, 6 if cpu_casts / best < 1e4 else 3), "unit": unit, "cores": cores, "kind": "port",
                   # the box: cores this process may run on / logical CPUs of the host; `cores` = threads the timed passes used
                   "affinity_cores": affinity, "host_cores": os.cpu_count(),
                   "value_1thread": round(c1ctr["rays"] / dt1 / 1e6, 6 if c1ctr["rays"] / dt1 < 1e4 else 3),
                   "sample": (f"{n} rays of this workload" if samp is None else f"every {n // len(samp)}th ray of this workload ({len(samp)} rays)")
                             + (f" x {B} casts ({casts} live)" if B > 1 else "")
                             + f", best of {reps} passes; 1 thread on {n1} rays. C restatement of {ref_name} (oracle/): an upper bound on the C# reference"}
        # parity check of the bench buffers themselves (not timed): every ray of this rank's shard, all eight fields, bit for bit --
        # BEFORE anything else is cast into them
        got = device_events(n)
        if samp is not None:
            got = got[samp]
        parity = bool(all(np.array_equal(got[f], ref[f]) for f in fields))
        parity_rays = len(got)            # how many rays the flag speaks for (a kd-tree on a large scene: a sample, see `samp`)
        del got
        bytes_pass = algorithmic_bytes(ctr, kind, links)
        pass_s = kern_ms * B * 1e-3                     # one pass of the hot path: B casts
        achieved = bytes_pass / pass_s / 1e9
        wkey = f"{scene}-{kind}-" + (f"D{domain}-" if kind == "voxel" else "") + f"n{n}" + (f"-b{B}" if B > 1 else "")
        prof = load_traffic(wkey)
        traffic = None if prof is None else prof.get("hbm_bytes_per_launch")
        # ---- roofline.own: the bytes THIS kernel's algorithm has to touch, counted by the counting build of the production kernel
        # itself (HARE_SHOOT_COUNT_OWN: the same launch geometry, the same events) on the same rays, once, after the timed region:
        #   104 B per cast (ray in, X_Event out) + 8 B per voxel walked into (64 B per tree node record fetched) + 4 B per list entry
        #   scanned + 32 B per candidate put through the FP32 pre-cull (its cull record) + 128 B per exact test (the polygon record)
        #   + 28 B per bounce.  The tight boxes, the pre-cull and the skipped empty leaves are IN this figure (the reference-priced
        #   `frac` charges 96 B for every intersect call of the reference, most of which the kernels never make), so it cannot pass 1
        #   unless the kernel moves fewer bytes than its own algorithm needs.
        own = None
        try:
            octr = torch.zeros(8, dtype=torch.int64, device="cuda")
            if B > 1:
                d_rays.copy_(d_rays0)
                part.bounce_device(n, d_rays.data_ptr(), B, d_excl.data_ptr(), d_events_last=d_out.data_ptr(), d_counters=octr.data_ptr(), stream=sp,
                                   flags=H.capi.SHOOT_COUNT_OWN)
            else:
                part.shoot_device(n, d_rays.data_ptr(), d_out.data_ptr(), d_counters=octr.data_ptr(), stream=sp, flags=H.capi.SHOOT_COUNT_OWN)
            torch.cuda.synchronize()
            oc = [int(x) for x in octr.cpu()]
            got_own = np.frombuffer(d_out[:n * 56].cpu().numpy().tobytes(), dtype=H.capi.XEVENT_DTYPE)
            if samp is not None:
                got_own = got_own[samp]
            own_parity = bool(all(np.array_equal(got_own[f], ref[f]) for f in fields))     # the counting build returns the same events
            del got_own
            if oc[0] == casts and own_parity:
                cw = 8 if kind == "voxel" else 64
                kb = 48 if bool((np.asarray(mesh.nverts) == 4).any()) else 32      # a topology with quadrilaterals: 48-byte cull records (CullFrame::stride)
                own_bytes = 104 * oc[0] + cw * oc[2] + 4 * oc[3] + kb * oc[5] + 128 * oc[4] + 28 * links
                own = {"frac": round(own_bytes / pass_s / 1e9 / HBM_PEAK_GBS, 4), "achieved": round(own_bytes / pass_s / 1e9, 1),
                       "bytes_per_launch": own_bytes // B, "bytes_per_cast": round(own_bytes / max(casts, 1), 1),
                       # S: walk operations the kernel EXECUTES per cast (counters word 6).  By default it equals C' (up to the drain's wide walk): round 6 made
                       # the step cheap (voxel_walk.h: 17 / 20 vector instructions where the compiler wrote 40) instead of skipping voxels; with the scene
                       # option "voxel_skip" -- the exact closed-form skip over empty 4^3 blocks -- S falls below C' and the kernel gets slower
                       "per_cast": {"C": round(oc[2] / max(casts, 1), 2), **({"S": round((oc[6] or oc[2]) / max(casts, 1), 2)} if kind == "voxel" else {}),
                                    "L": round(oc[3] / max(casts, 1), 2),
                                    "K": round(oc[5] / max(casts, 1), 2), "T": round(oc[4] / max(casts, 1), 2)},
                       "formula": f"104+{cw}C'+4L'+{kb}K'+128T' (+28/bounce), counted by " + part.kernel_name(n, flags=H.capi.SHOOT_COUNT_OWN)}
        except Exception as e:       # a batch whose kernel has no counting build (HARE_E_UNSUPPORTED): the line says so
            own = {"frac": None, "why": str(e)[:120]}
        # ---- roofline.issue: the bound the counters name (VALU issue), as a fraction: PMC wave-instruction counts by class x the measured
        # cost of a class on a SIMD (VALU_NS), over the SIMDs' time in the kernel.  A LOWER bound on how busy the vector issue is: FP64
        # compares, selects and conversions are priced as "other", and an instruction with idle lanes costs what a full one does.
        issue = None
        if prof is not None and prof.get("SQ_INSTS_VALU"):
            valu = float(prof["SQ_INSTS_VALU"])
            f64 = {k: float(prof.get(k2) or 0) for k, k2 in (("add_f64", "SQ_INSTS_VALU_ADD_F64"), ("mul_f64", "SQ_INSTS_VALU_MUL_F64"),
                                                            ("fma_f64", "SQ_INSTS_VALU_FMA_F64"), ("trans_f64", "SQ_INSTS_VALU_TRANS_F64"))}
            have_f64 = prof.get("SQ_INSTS_VALU_FMA_F64") is not None
            ns = sum(f64[k] * VALU_NS[k] for k in f64) + max(0.0, valu - sum(f64.values())) * VALU_NS["other"]
            issue = {"frac": round(ns / N_SIMDS / (kern_ms * 1e6), 4), "valu_insts": int(valu),
                     "f64_insts": int(sum(f64.values())) if have_f64 else None, "lane_util": prof.get("valu_lane_util"),
                     "wait_frac": prof.get("sq_wait_any_frac"), "ta_busy": prof.get("ta_busy_frac"),
                     "l1_per_ray": prof.get("l1_accesses_per_ray")}
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "kernel": kernel_name, "kernel_ms": round(kern_ms, 4), "kernel_ms_source": kern_src,
                    "kernel_ms_event_pair_per_launch": round(kern_ms_pairs, 4),
                    "algorithmic_bytes_per_launch": bytes_pass // B,
                    "bytes_per_cast": round(bytes_pass / max(casts, 1), 1),
                    "per_cast": {"C": round(ctr["cells"] / max(casts, 1), 2), "L": round(ctr["entries"] / max(casts, 1), 2),
                                 "T": round(ctr["tests"] / max(casts, 1), 2)},
                    # `frac` prices the REFERENCE algorithm's cells / list entries / tests per ray (counted by the oracle):
                    # a rate in units of the reference's work.  `own.frac` prices what the kernel itself touches; `issue.frac` is the
                    # fraction of the bound the counters name.
                    "own": own, "issue": issue, "measured_bound": "valu issue + dependent-load latency (not HBM)",
                    "hbm_busy_frac": None if traffic is None else round(traffic / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "device_copy_gbs": None if copy_gbs is None else round(copy_gbs, 1)}
        try:
            if kind == "voxel" and B == 1 and n >= 1572864 and part.get_option("voxel_order") == 1:
                # a large batch of primary rays: a step is the order pass + the pool kernel; kernel_ms is the step's
                roofline["step_kernels"] = "hare_cost_order (~12 us per million rays) + " + kernel_name
        except Exception:
            pass
        if roofline["frac"] > 1.0:
            roofline["warning"] = "frac > 1: reference-priced numerator (SURVEY 8(d)), not a hardware fraction; read own.frac / issue.frac"
        if B > 1:
            roofline["bounce_loop_ms"] = round(loop_ms, 4)
            roofline["bounce_loop"] = "hare_bounce_device: " + (part.bounce_kernel_name(n, B) or "a launch per cast")
            roofline["live_casts_per_pass"] = casts
        if bounce_batch is not None:
            bb_ev = bounce_batch.pop("events")
            bounce_batch["parity_vs_oracle"] = bool(all(np.array_equal(bb_ev[f], ref[f]) for f in fields)) and bounce_batch["casts"] == casts
    if bounce_batch is not None:
        bounce_batch.pop("events", None)
    if w["cpu_baseline"] and dist is not None:
        # one flag per rank, gathered: the line's parity is the AND over every device
        fl = torch.tensor([1 if parity else 0], dtype=torch.int64)
        if backend == "nccl":
            fl = fl.cuda()
        got_fl = [torch.zeros_like(fl) for _ in range(world)]
        dist.all_gather(got_fl, fl)
        parity_ranks = [bool(int(g[0])) for g in got_fl]
    if rank != 0:
        del ray_sets, out_sets, d_rays, d_out, d_rays0, d_excl
        torch.cuda.empty_cache()
        return None

    ms_per_step = wall * 1e3 / steps
    value = n_total * B * steps / wall / 1e6   # casts per second
    line = {
        "metric": "Mrays/s (primary hits) into 100k-tri mesh", "value": round(value, 2), "unit": "Mrays/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": (f"{n_total} spherical-Fibonacci burst rays in contiguous shards over {world} GPU(s)" if strong else
                                f"{n} spherical-Fibonacci burst rays per GPU") + f" -> {mesh.name} "
                               f"({mesh.P} triangles), {kdesc}, closest hit (X_Event)"
                               + (f", x{B} specular bounces device-resident (value = casts/s)" if B > 1 else ""),
                   "workload_short": f"{n_total} burst rays -> {mesh.name}, {kdesc}" + (f", x{B} specular bounces" if B > 1 else ""),
                   "rays_per_gpu": n, "rays_total": n_total, "triangles": mesh.P, "partition": kdesc,
                   "sharding": f"rays x{world}, scene replicated",
                   "buffer_sets": n_sets,
                   "backend": ("none" if dist is None else ("rccl" if backend == "nccl" else "gloo (rehearsal)"))},
        "ms_per_step_per_rank": {"max": round(max(walls) * 1e3 / steps, 4), "min": round(min(walls) * 1e3 / steps, 4)},
        "device_ms_per_step": round(dev_ms / steps, 4), "kernel_only_mrays_s": round(n / kern_ms / 1e3, 2),
        "end_to_end_mrays_s": None if e2e is None else round(e2e, 1),
        "hits": hits_total, "rays": rays_total, "build_s": round(build_s, 3),
        "x_event_parity_vs_oracle": parity if parity_ranks is None else bool(all(parity_ranks)),
        "roofline": roofline, "cpu_baseline": cpu,
    }
    if parity_rays is not None and parity_rays != n:
        line["parity_rays_checked"] = parity_rays          # fewer than the batch: the oracle got a stated sample (cpu_baseline.sample)
    if dist is not None:
        line["ranks_seen_in_reduce"] = ranks_seen        # from the all-reduced ray counter (RCCL at backend nccl)
        if parity_ranks is not None:
            line["parity_per_rank"] = parity_ranks
            line["parity_sample"] = (f"rank 0: its whole shard ({n} rays" + (f" x {B} casts" if B > 1 else "") + "), all 8 X_Event fields "
                                     f"bit-equal to the oracle; ranks 1..{world - 1}: the first {min(n, PARITY_SAMPLE_RAYS)} rays of their shard")
    if two_streams is not None:
        line["two_streams"] = two_streams
    if e2e_slim is not None:
        line["end_to_end_slim_mrays_s"] = round(e2e_slim, 1)
        line["slim_events_rebuild_identical"] = slim_ok
    if bounce_batch is not None:
        line["bounce_batch"] = bounce_batch
    del ray_sets, out_sets, d_rays, d_out, d_rays0, d_excl
    torch.cuda.empty_cache()
    return line


PARITY_SAMPLE_RAYS = 65536      # what ranks 1..N-1 pass through the oracle (rank 0: its whole shard)

# What the default run appends to the headline line: the rest of BASELINE.json's table.
#   c4 / c5     configs 4 and 5 as BASELINE states them -- ONE batch of 16 777 216 rays (8 388 608 rays x 8 specular bounces) into the
#               1M-triangle cathedral, cut into contiguous shards over the N ranks (strong scaling; at N = 8: 2M / 1M rays per GPU).
#               Measured at EVERY N, with the counter all-reduce, max-over-ranks timing and a parity flag per rank.
#   c3, c4_shard, c5_shard (N = 1 only): config 3, and what ONE GPU of the 8-GPU configs gets (2M rays; 1M rays x 8), with roofline
#               and the CPU baseline -- the per-GPU kernels' figures, comparable round to round.
EXTRA_CONFIGS = (
    ("c3", {"scene": "hall", "kind": "octree", "domain": 64, "rays": 1 << 20, "bounces": 1, "steps": 5, "warmup": 1, "only_n1": True,
            "two_streams": True}),
    ("c2_quads", {"scene": "hall_quads", "kind": "voxel", "domain": 64, "rays": 1 << 20, "bounces": 1, "steps": 8, "warmup": 2, "only_n1": True}),
    ("c4_shard", {"scene": "cathedral", "kind": "voxel", "domain": 128, "rays": 1 << 21, "bounces": 1, "steps": 8, "warmup": 2,
                  "only_n1": True}),
    ("c5_shard", {"scene": "cathedral", "kind": "voxel", "domain": 128, "rays": 1 << 20, "bounces": 8, "steps": 3, "warmup": 1,
                  "only_n1": True, "bounce_api": "batch"}),
    ("c4", {"scene": "cathedral", "kind": "voxel", "domain": 128, "rays_total": 1 << 24, "bounces": 1, "steps": 5, "warmup": 1}),
    ("c5", {"scene": "cathedral", "kind": "voxel", "domain": 128, "rays_total": 1 << 23, "bounces": 8, "steps": 2, "warmup": 1}),
)
LINE_BUDGET = 7400      # the driver keeps the last 8 KB of stdout: the ONE line, with every config in it, must fit (round 4's was 12.5 KB
                        # and lost "c3" to the cut)


def compact_sub(sub: dict) -> dict:
    """An appended config as it goes into the line: the numbers, the kernel, the three fractions, parity -- no prose (the headline
    entry carries the explanations once)."""
    rf = sub.get("roofline") or {}
    own, issue, cpu = rf.get("own") or {}, rf.get("issue") or {}, sub.get("cpu_baseline")
    out = {k: sub[k] for k in ("value", "unit", "n_gpus", "scaling", "steps", "warmup", "ms_per_step", "kernel_only_mrays_s", "hits", "rays",
                               "x_event_parity_vs_oracle", "parity_rays_checked", "parity_per_rank", "ranks_seen_in_reduce") if k in sub}
    out["config"] = {k: sub["config"][k] for k in ("rays_per_gpu", "rays_total", "backend")}
    out["config"]["workload"] = sub["config"]["workload_short"]
    r = {k: rf.get(k) for k in ("frac", "achieved", "traffic", "kernel", "kernel_ms", "algorithmic_bytes_per_launch", "bytes_per_cast", "per_cast")}
    if rf.get("warning"):
        r["warning"] = "frac > 1: reference-priced; see own / issue"
    r["own"] = {k: own.get(k) for k in ("frac", "bytes_per_cast", "per_cast", "why") if own.get(k) is not None} if own else None
    r["issue"] = {k: issue.get(k) for k in ("frac", "lane_util")} if issue else None
    for k in ("live_casts_per_pass", "bounce_loop_ms", "step_kernels"):
        if k in rf:
            r[k] = rf[k]
    out["roofline"] = r
    out["cpu_baseline"] = None if not cpu else {k: cpu[k] for k in ("value", "unit", "cores", "kind", "host_cores", "value_1thread") if k in cpu}
    if "bounce_batch" in sub:
        out["bounce_batch"] = sub["bounce_batch"]
    if "two_streams" in sub:
        out["two_streams"] = {k: sub["two_streams"][k] for k in ("value", "ms_per_step")}
    return out


def fit_line(line: dict) -> str:
    """The one JSON line, within LINE_BUDGET: what is dropped first is prose, never a number."""
    dumps = lambda o: json.dumps(o, separators=(",", ":"))
    txt = dumps(line)
    line.get("config", {}).pop("workload_short", None)
    drops = [("cpu_baseline", "sample"), ("roofline", "measured_bound"), ("parity_sample",), ("config", "sharding"), ("config", "partition"),
             ("roofline", "own", "formula"), ("roofline", "warning"), ("two_streams", "note"),
             ("inproc_sharded", "workload"), ("inproc_sharded", "parity_sample")]
    for path in drops:
        if len(txt) <= LINE_BUDGET:
            break
        d = line
        for k in path[:-1]:
            d = d.get(k) or {}
        if isinstance(d, dict):
            d.pop(path[-1], None)
        txt = dumps(line)
    for name in list((line.get("configs") or {}).keys()):
        if len(txt) <= LINE_BUDGET:
            break
        line["configs"][name]["config"].pop("workload", None)
        line["configs"][name]["roofline"].pop("warning", None)
        txt = dumps(line)
    return txt


def inproc_sharded(env, n_scenes: int, n_total: int, domain: int = 64, bounces: int = 8) -> dict:
    """The path a ONE-process caller takes on a multi-GPU node -- what Pachyderm's C# host is (Spatial_Partition.cs:27-35: one object, one
    caller): ONE hare_shoot_batch_sharded and ONE hare_bounce_batch_sharded call from host buffers over `n_scenes` equal scenes, scene k on
    device k % device_count, contiguous ray shards, a host thread per scene inside the library (api.cpp, bounce.cpp).  The N ranks of this
    bench measure the kernels; this leg measures that call: PCIe-inclusive (host rays in, host X_Events out), so its rate is the host
    link's, not the kernels'.  Parity: the concatenated events against the oracle on every `stride`-th ray of the whole batch (so every
    shard, i.e. every device, is sampled), all eight fields, both calls.  Rank 0 only, after the timed regions."""
    import numpy as np

    import hare_amd as H
    from oracle import pyoracle as po

    ndev = max(1, H.device_count())
    mesh, topo = env.mesh(H, "hall")
    devs = [k % ndev for k in range(n_scenes)]
    t0 = time.time()
    parts = [H.Voxel_Grid([topo], domain, device=d) for d in devs]
    build_s = time.time() - t0
    rays = H.scenes.burst_rays(n_total, mesh.size)
    SP = H.Spatial_Partition
    fields = ("hit", "poly_id", "t", "x", "y", "z", "u", "v")
    out = {"scenes": n_scenes, "devices": devs, "device_count": ndev, "rays_total": n_total,
           "workload": f"{n_total} burst rays -> {mesh.name}, Voxel_Grid Domain={domain}, host buffers, one call",
           "kernels": sorted({p.kernel_name(max(1, n_total // n_scenes)) for p in parts}), "build_s": round(build_s, 2)}
    try:
        # The C-ABI calls themselves, on buffers the caller already holds (as the e2e leg times hare_shoot_batch, and as a C# caller with
        # its arrays pinned does): the Python wrappers allocate a fresh 56-byte-per-ray result array per call, whose first-touch page
        # faults are the wrapper's cost, not the library's (1M rays: 13.6 ms through the wrapper, round 6's first run of this leg).
        import ctypes as C
        lib, kind = H.capi.lib, parts[0]._kind
        handles = (C.c_void_p * n_scenes)(*[p._h for p in parts])
        rays = np.ascontiguousarray(rays, np.float64)
        ev, evb = np.zeros(n_total, H.capi.XEVENT_DTYPE), np.zeros(n_total, H.capi.XEVENT_DTYPE)
        c1, c2 = H.capi.Counters(), H.capi.Counters()

        def shoot_call():
            H.capi.check(lib.hare_shoot_batch_sharded(handles, n_scenes, kind, 0, n_total, rays.ctypes.data, None, None, 0, ev.ctypes.data, C.addressof(c1)))

        def bounce_call():
            H.capi.check(lib.hare_bounce_batch_sharded(handles, n_scenes, kind, 0, n_total, rays.ctypes.data, None, None, bounces, 0, None,
                                                       evb.ctypes.data, C.addressof(c2), None))

        shoot_call()                                        # sizes every scene's staging buffers, touches `ev`: not part of the measurement
        best = None
        for _ in range(3):
            t1 = time.perf_counter()
            shoot_call()
            dt = time.perf_counter() - t1
            best = dt if best is None else min(best, dt)
        ctr = c1.as_dict()
        out["shoot"] = {"mrays_s": round(n_total / best / 1e6, 1), "ms": round(best * 1e3, 3), "hits": ctr["hits"], "rays": ctr["rays"]}
        bounce_call()
        best = None
        for _ in range(2):
            t1 = time.perf_counter()
            bounce_call()
            dt = time.perf_counter() - t1
            best = dt if best is None else min(best, dt)
        ctrb = c2.as_dict()
        out["bounce"] = {"mcasts_s": round(ctrb["rays"] / best / 1e6, 1), "ms": round(best * 1e3, 3), "casts": ctrb["rays"], "bounces": bounces}
        out["timed"] = "the C-ABI call on buffers the caller holds (host rays in, host X_Events out)"
        # the checker, on a sample that touches every shard
        stride = max(1, n_total // 65536)
        idx = np.arange(0, n_total, stride)
        ot, og, _ = env.oracle(H, "hall", "voxel", domain)
        nt = min(len(os.sched_getaffinity(0)), 64)
        r = np.ascontiguousarray(rays[idx])
        ref, _ = og.shoot(r, nthreads=nt)
        ok1 = all(np.array_equal(ev[f][idx], ref[f]) for f in fields)
        excl = None
        cur = r
        for b in range(bounces):
            if excl is None:
                e, _ = og.shoot(cur, nthreads=nt)
            else:
                live = excl >= 0
                e = np.zeros(len(cur), po.XEVENT_DTYPE)
                e["poly_id"] = -1
                if live.any():
                    e[live], _ = og.shoot(cur[live], excl1=excl[live], nthreads=nt)
            if b + 1 < bounces:
                cur = po.reflect_batch(ot, cur, e)
                excl = np.where(e["hit"] != 0, e["poly_id"], -2).astype(np.int32)
        okb = all(np.array_equal(evb[f][idx], e[f]) for f in fields)
        out["parity_vs_oracle"] = bool(ok1 and okb)
        out["parity_sample"] = f"every {stride}th ray of the batch ({len(idx)} rays, every shard), 8 fields: the shoot's events and the last cast's"
    except Exception as ex:                                   # the leg must never take the line down with it
        out["error"] = str(ex)[:200]
    for p in parts:
        p.close()
    return out


def main() -> None:
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # the driver's command line: start the ranks ourselves, BEFORE anything initialises the GPU in this process
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: one rank per GPU, the two must agree")

    # dmabuf IPC is the only kind this pool's host driver supports (RCCL over xGMI needs it); exported on the boxes already, and set here
    # for a launcher that dropped it -- before anything initialises HIP
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the ray-cast path has no CPU fallback")
    ndev = torch.cuda.device_count()
    if world > ndev and args.backend == "nccl":
        raise SystemExit(f"--gpus {world} with backend nccl needs {world} GPUs, {ndev} visible "
                         f"(--backend gloo rehearses the multi-rank path with ranks sharing a GPU)")
    device = local_rank % ndev
    torch.cuda.set_device(device)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist  # noqa: F811
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:                    # --force-dist on one rank: a process group of one
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            # RCCL prints a version banner on STDOUT when it builds its first communicator: the contract is ONE JSON line there, so the
            # process's fd 1 points at stderr while that happens (a C library's printf does not go through sys.stdout)
            sys.stdout.flush()
            _fd1 = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device))
                # RCCL builds its communicator on the first collective: do that here, never inside the timed region
                _t = torch.zeros(8, dtype=torch.int64, device="cuda")
                dist.all_reduce(_t)
                torch.cuda.synchronize()
            finally:
                try:
                    import ctypes
                    ctypes.CDLL(None).fflush(None)          # whatever the C side still holds in its stdio buffer goes where fd 1 points NOW
                except Exception:
                    pass
                os.dup2(_fd1, 1)
                os.close(_fd1)
        else:
            dist.init_process_group(backend="gloo")
            dist.all_reduce(torch.zeros(8, dtype=torch.int64))

    env = Env(args, rank, local_rank, world, dist, device)
    head = {"scene": args.scene, "kind": args.kind, "domain": args.domain, "rays": args.rays, "bounces": args.bounces,
            "steps": args.steps, "warmup": args.warmup, "cpu_baseline": not args.no_cpu_baseline, "e2e": not args.no_e2e,
            "bounce_api": args.bounce_api, "two_streams": True}
    line = measure(head, env)

    default_workload = (args.scene == "hall" and args.kind == "voxel" and args.domain == 64 and args.rays == 1 << 20 and args.bounces == 1)
    extras = (default_workload and not args.no_extra_configs) or args.extra_configs
    if extras and not args.no_cpu_baseline:
        # the rest of BASELINE's table in the same process(es), driver-observed like the headline; fewer steps, one oracle pass.
        # Every rank takes part: the strong-scaled configs 4 and 5 shard over all of them
        configs = {}
        prev_scene = args.scene
        for name, cfg in EXTRA_CONFIGS:
            if cfg.get("only_n1") and world > 1:
                continue
            if cfg["scene"] != prev_scene:
                env.drop(prev_scene)
                prev_scene = cfg["scene"]
            cfg = {k: v for k, v in cfg.items() if k != "only_n1"}
            if args.extra_rays > 0:
                if "rays_total" in cfg:
                    cfg["rays_total"] = min(cfg["rays_total"], args.extra_rays * world)
                else:
                    cfg["rays"] = min(cfg["rays"], args.extra_rays)
            w = dict({"bounce_api": "device"}, **cfg, cpu_baseline=True, e2e=False, copy_bw=False, cpu_reps=2, cpu_budget_s=6.0)
            if "rays_total" in cfg:
                w["cpu_reps"] = 1
            t0 = time.time()
            sub = measure(w, env)
            if rank == 0:
                sub = compact_sub(sub)
                sub["wall_s"] = round(time.time() - t0, 1)
                configs[name] = sub
        if rank == 0:
            line["configs"] = configs
    # the single-process multi-GPU calls (hare_*_batch_sharded), on rank 0 while the other ranks wait at the final barrier
    k_scenes = args.inproc_scenes
    if k_scenes < 0:
        k_scenes = ndev if world > 1 else 0
    if rank == 0 and k_scenes > 0 and not args.no_cpu_baseline:
        line["inproc_sharded"] = inproc_sharded(env, k_scenes, args.rays * world, args.domain if args.kind == "voxel" else 64)
    if rank == 0:
        print(fit_line(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
