#!/usr/bin/env python3
"""bench.py -- Mrays/s of the Hare ray-cast hot path on MI355X (BASELINE.json metric).

A step = one pass of the hot path over one batch: `rays_per_gpu` primary rays (spherical-Fibonacci
burst) cast into the 100k-triangle hall through Voxel_Grid's 3D-DDA kernel (BASELINE.json
configs[1]), inputs and outputs resident in HBM.  With N GPUs the burst is sharded contiguously
(rank r casts rays [r*n, (r+1)*n) of an N*n-ray burst: weak scaling, scene replicated, no data-path
collective); the only exchange is the RCCL all-reduce of the hit counter after each step.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  `roofline.achieved` = algorithmic bytes per launch (SURVEY.md 8(d):
104 + 8*C + 4*L + 96*T per ray, C/L/T counted exactly by the oracle on the same rays) / average
kernel duration measured here with HIP events on the launch stream.  `cpu_baseline` = the oracle
(C restatement of Voxel_Grid.Shoot, kind "port") timed on this box's host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes(ctr: dict) -> int:
    """SURVEY.md 8(d) / BASELINE.md 4: B = sum over rays of 104 + 8*C + 4*L + 96*T."""
    return 104 * ctr["rays"] + 8 * ctr["cells"] + 4 * ctr["entries"] + 96 * ctr["tests"]


def load_traffic(workload_key: str):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json), or None."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(p) as f:
            t = json.load(f)
        e = t.get(workload_key)
        return None if e is None else e.get("hbm_bytes_per_launch")
    except Exception:
        return None


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rays", type=int, default=1 << 20, help="rays per GPU per step")
    ap.add_argument("--domain", type=int, default=64, help="Voxel_Grid Domain")
    ap.add_argument("--scene", default="hall", choices=["hall", "cathedral", "shoebox"])
    ap.add_argument("--kind", default="voxel", choices=["voxel", "octree", "kdtree"])
    ap.add_argument("--bounces", type=int, default=1,
                    help="casts per step: >1 = device-resident specular bounce loop (BASELINE config 5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse on one GPU)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the ray-cast path has no CPU fallback")
    device = local_rank % torch.cuda.device_count()   # one GPU per rank on a real node; shared only in a gloo rehearsal
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist  # noqa: F811
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend="gloo")
        # RCCL builds its communicator on the first collective: do that here, never inside the timed region (--warmup 0)
        _t = torch.zeros(8, dtype=torch.int64, device="cuda")
        dist.all_reduce(_t)
        torch.cuda.synchronize()

    import hare_amd as H
    from hare_amd.sharding import shard_range

    mesh = H.scenes.SCENES[args.scene]()
    topo = H.Topology(mesh.verts, mesh.nverts)
    t0 = time.time()
    if args.kind == "voxel":
        part = H.Voxel_Grid([topo], args.domain, device=device)
        kdesc = f"Voxel_Grid Domain={args.domain}"
    elif args.kind == "octree":
        part = H.Octree([topo], 8, 16, device=device)
        kdesc = "Octree maxDepth=8 maxPolys=16"
    else:
        part = H.KDTree([topo], 12, 16, device=device)
        kdesc = "KDTree maxDepth=12 maxPolys=16"
    build_s = time.time() - t0
    kernel_name = {"voxel": "hare_voxel_persist_tri", "octree": "hare_octree_persist", "kdtree": "hare_kdtree_shoot"}[args.kind]

    n = args.rays
    n_total = n * world
    lo, hi = shard_range(n_total, rank, world)
    rays_h = H.scenes.burst_rays(n_total, mesh.size, start=lo, count=hi - lo)
    d_rays = torch.from_numpy(rays_h).cuda()
    d_out = torch.empty(n * 56, dtype=torch.uint8, device="cuda")
    d_ctr = torch.zeros(8, dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream()

    d_rays0 = d_rays.clone() if args.bounces > 1 else None
    d_excl = torch.full((n,), -1, dtype=torch.int32, device="cuda") if args.bounces > 1 else None
    # the per-batch hit-count reduce runs on RCCL's stream, overlapped with the NEXT batch's kernel:
    # two counter blocks alternate, a block is reused only after its all-reduce has been waited for
    ctrs = [d_ctr, torch.zeros_like(d_ctr)]
    pending = [None, None]
    state = {"k": 0}

    def step():
        k = state["k"]
        state["k"] = k + 1
        c = ctrs[k & 1]
        if pending[k & 1] is not None:
            pending[k & 1].wait()
            pending[k & 1] = None
        c.zero_()
        if args.bounces > 1:     # config 5: shoot -> reflect -> shoot with poly_origin1 = the polygon just hit
            d_rays.copy_(d_rays0)
            d_excl.fill_(-1)
            for b in range(args.bounces):
                part.shoot_device(n, d_rays.data_ptr(), d_out.data_ptr(), d_excl1=d_excl.data_ptr(),
                                  d_counters=c.data_ptr(), stream=stream.cuda_stream)
                if b + 1 < args.bounces:
                    part.reflect_device(n, d_rays.data_ptr(), d_out.data_ptr(), d_excl.data_ptr(), stream=stream.cuda_stream)
        else:
            part.shoot_device(n, d_rays.data_ptr(), d_out.data_ptr(), d_counters=c.data_ptr(),
                              stream=stream.cuda_stream)
        if dist is not None:
            pending[k & 1] = dist.all_reduce(c, async_op=True)   # RCCL: the final hit-count reduce (64 B)

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
    part.shoot_device(n, d_rays.data_ptr(), d_out.data_ptr(), stream=stream.cuda_stream)
    if d_rays0 is not None:
        d_rays.copy_(d_rays0)
    for _ in range(args.warmup):
        step()
    fence()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t_start = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    fence()
    wall = time.perf_counter() - t_start
    dev_ms = ev0.elapsed_time(ev1)
    t = torch.tensor([wall, dev_ms], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall, dev_ms = float(t[0]), float(t[1])
    last = ctrs[(state["k"] - 1) & 1]
    hits_total = int(last[1])
    rays_total = int(last[0])

    # kernel-only duration: K launches back to back on the launch stream, HIP events around them
    kern_ms = None
    if True:
        for _ in range(2):
            part.shoot_device(n, d_rays.data_ptr(), d_out.data_ptr(), stream=stream.cuda_stream)
        torch.cuda.synchronize()
        k0 = torch.cuda.Event(enable_timing=True)
        k1 = torch.cuda.Event(enable_timing=True)
        k0.record(stream)
        for _ in range(args.steps):
            part.shoot_device(n, d_rays.data_ptr(), d_out.data_ptr(), stream=stream.cuda_stream)
        k1.record(stream)
        torch.cuda.synchronize()
        kern_ms = k0.elapsed_time(k1) / args.steps

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- rank 0: oracle pass over THIS rank's rays = exact C/L/T for the roofline + the CPU baseline
    roofline = None
    cpu = None
    parity = None
    if not args.no_cpu_baseline and args.kind == "voxel" and args.bounces == 1:
        from oracle import pyoracle as po
        cores = int(os.environ.get("HARE_CPU_THREADS", "0")) or min(len(os.sched_getaffinity(0)), 32)
        ot = po.Topology(mesh.verts, mesh.nverts)
        og = po.VoxelGrid([ot], domain=args.domain)
        best = None
        ctr = None
        ref = None
        budget = time.time() + 20.0
        reps = 0
        max_reps = 5 if world == 1 else 1   # the CPU baseline is reported at N = 1 only
        while reps < max_reps and (reps < 1 or time.time() < budget):
            c0 = time.perf_counter()
            ref, ctr = og.shoot(rays_h, nthreads=cores)
            dt = time.perf_counter() - c0
            best = dt if best is None else min(best, dt)
            reps += 1
        n1 = min(n, 100000)
        c0 = time.perf_counter()
        og.shoot(rays_h[:n1], nthreads=1)
        dt1 = time.perf_counter() - c0
        cpu = None if world > 1 else {"value": round(n / best / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
               "sample": f"{n} rays of this workload (rank 0's shard), best of {reps} passes on {cores} threads; "
                         f"1 thread: {n1 / dt1 / 1e6:.3f} Mrays/s on {n1} rays. C restatement of Hare "
                         f"Voxel_Grid.Shoot (oracle/, per-thread mailbox, no per-candidate allocation): an upper "
                         f"bound on the C# reference, which cannot be run here"}
        bytes_launch = algorithmic_bytes(ctr)
        achieved = bytes_launch / (kern_ms * 1e-3) / 1e9
        wkey = f"{args.scene}-{args.kind}-D{args.domain}-n{n}"
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": load_traffic(wkey),
                    "kernel": kernel_name, "kernel_ms": round(kern_ms, 4),
                    "algorithmic_bytes_per_launch": bytes_launch,
                    "bytes_per_ray": round(bytes_launch / n, 1),
                    "per_ray": {"C_cells": round(ctr["cells"] / n, 2), "L_entries": round(ctr["entries"] / n, 2),
                                "T_tests": round(ctr["tests"] / n, 2)}}
        # parity spot check of the bench buffers themselves (not timed)
        got = np.frombuffer(d_out.cpu().numpy().tobytes(), dtype=H.capi.XEVENT_DTYPE)
        parity = bool(all(np.array_equal(got[f], ref[f]) for f in ("hit", "poly_id", "t", "x", "y", "z")))

    ms_per_step = wall * 1e3 / args.steps
    value = n_total * args.bounces * args.steps / wall / 1e6   # casts per second
    line = {
        "metric": "Mrays/s (primary hits) into 100k-tri mesh", "value": round(value, 2), "unit": "Mrays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{n} spherical-Fibonacci burst rays per GPU -> {mesh.name} "
                               f"({mesh.P} triangles), {kdesc}, closest hit (X_Event)"
                               + (f", x{args.bounces} specular bounces device-resident (value = casts/s)" if args.bounces > 1 else ""),
                   "rays_per_gpu": n, "triangles": mesh.P, "partition": kdesc, "sharding": f"rays x{world}, scene replicated"},
        "device_ms_per_step": round(dev_ms / args.steps, 4), "kernel_only_mrays_s": round(n / kern_ms / 1e3, 2),
        "hits": hits_total, "rays": rays_total, "build_s": round(build_s, 3),
        "x_event_parity_vs_oracle": parity,
        "roofline": roofline, "cpu_baseline": cpu,
    }
    print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
