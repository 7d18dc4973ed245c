"""bench.py on the GPU box: the driver's command lines at small sizes.  `--gpus 2 --backend gloo` runs the real
two-rank path (two processes, each with its own scene and HIP shooter, sharing this box's one GPU; the hit counter
is all-reduced across them) -- the nccl backend needs two GPUs, which the driver's 8-GPU node has."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import hare_amd as H

pytestmark = pytest.mark.gpu
OCTREE_KERNEL = "hare_octree_group"      # what hare_shoot_kernel_name reports for Octree.Shoot batches
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    # ONE JSON line and nothing else on stdout: RCCL's version banner (printed by the C library when the first communicator is built)
    # goes to stderr (bench.py points fd 1 there while the process group comes up)
    assert [ln for ln in r.stdout.splitlines() if ln.strip()] == lines, r.stdout[:2000]
    assert len(lines[0]) < 8000, len(lines[0])         # the driver keeps the last 8 KB of stdout: the whole line must survive
    return json.loads(lines[0])


def _oracle_hits(n_total, scene="hall", domain=64, bounces=1):
    """Hits summed over `bounces` casts of the n_total-ray burst (the oracle's own bounce loop: reflect, exclude the polygon left)."""
    from oracle import pyoracle as po
    mesh = H.scenes.SCENES[scene]()
    rays = H.scenes.burst_rays(n_total, mesh.size)
    ot = po.Topology(mesh.verts, mesh.nverts)
    og = po.VoxelGrid([ot], domain=domain)
    hits, excl = 0, None
    for b in range(bounces):
        if excl is None:
            ev, _ = og.shoot(rays, nthreads=8)
        else:
            live = excl >= 0
            ev = np.zeros(len(rays), po.XEVENT_DTYPE)
            ev["poly_id"] = -1
            if live.any():
                ev[live], _ = og.shoot(rays[live], excl1=excl[live], nthreads=8)
        hits += int(ev["hit"].sum())
        if b + 1 < bounces:
            rays = po.reflect_batch(ot, rays, ev)
            excl = np.where(ev["hit"] != 0, ev["poly_id"], -2).astype(np.int32)
    return hits


def test_bench_two_ranks_started_by_bench_itself():
    n = 65536
    j = _bench("--gpus", "2", "--backend", "gloo", "--rays", str(n), "--steps", "3", "--warmup", "1")
    assert j["n_gpus"] == 2 and j["scaling"] == "weak"
    assert j["rays"] == 2 * n                                   # both ranks' counters reached the reduce
    assert j["hits"] == _oracle_hits(2 * n)                     # ... and they cast the two halves of ONE 2n-ray burst
    assert j["x_event_parity_vs_oracle"] is True                # rank 0's shard, bit for bit
    assert j["ms_per_step_per_rank"]["max"] >= j["ms_per_step_per_rank"]["min"] > 0
    assert j["cpu_baseline"] is None                            # reported at N = 1 only
    assert j["roofline"]["kernel"] == "hare_voxel_pool_tri" and j["roofline"]["frac"] > 0
    assert j["ranks_seen_in_reduce"] == 2 and j["parity_per_rank"] == [True, True]


def test_bench_two_ranks_print_configs_4_and_5_strong_scaled():
    """The driver's `python bench.py --gpus N` must measure BASELINE's 8-GPU configs, not only the headline: at N > 1 the line carries
    c4 (ONE burst into the cathedral, D = 128, cut over the ranks) and c5 (the same with 8 specular bounces), each reduced over both
    ranks and checked against the oracle on both.  Here at 2 x 32768 rays, gloo, the two ranks sharing this box's GPU."""
    per = 32768
    j = _bench("--gpus", "2", "--backend", "gloo", "--rays", str(per), "--steps", "2", "--warmup", "1", "--extra-configs",
               "--extra-rays", str(per))
    assert set(j["configs"]) == {"c4", "c5"}
    c4, c5 = j["configs"]["c4"], j["configs"]["c5"]
    for c in (c4, c5):
        assert c["n_gpus"] == 2 and c["scaling"] == "strong" and c["config"]["rays_total"] == 2 * per
        assert c["config"]["rays_per_gpu"] == per and c["ranks_seen_in_reduce"] == 2
        assert c["x_event_parity_vs_oracle"] is True and c["parity_per_rank"] == [True, True]
        assert c["roofline"]["kernel"] == "hare_voxel_pool_tri_g" and c["roofline"]["frac"] > 0 and c["value"] > 0
    assert c4["rays"] == 2 * per                                      # the full config's ray count reached the reduce
    assert c4["hits"] == _oracle_hits(2 * per, "cathedral", 128)
    assert "x8 specular bounces" in c5["config"]["workload"]
    assert c5["hits"] == _oracle_hits(2 * per, "cathedral", 128, bounces=8) and c5["rays"] <= 8 * 2 * per
    assert c5["roofline"]["live_casts_per_pass"] > per * 7           # rank 0's shard, from its oracle pass


def test_bench_rccl_branch_on_one_gpu_with_a_process_group_of_one():
    """The driver's 8-GPU run is the first time ranks > 1 meet RCCL; its code path -- init_process_group(nccl, device_id), the
    first all-reduce, the per-step async all-reduce with the two-slot hand-off, the all-gather of CUDA tensors, barrier and
    destroy -- runs here with a process group of one and must give the line the plain run gives."""
    n = 65536
    a = _bench("--rays", str(n), "--steps", "4", "--warmup", "2")
    b = _bench("--rays", str(n), "--steps", "4", "--warmup", "2", "--force-dist", "--backend", "nccl")
    assert a["config"]["backend"] == "none" and b["config"]["backend"] == "rccl"
    assert b["n_gpus"] == 1 and (b["hits"], b["rays"]) == (a["hits"], a["rays"]) and b["rays"] == n
    assert a["x_event_parity_vs_oracle"] is True and b["x_event_parity_vs_oracle"] is True
    assert b["cpu_baseline"]["value"] > 0 and b["roofline"]["frac"] > 0
    assert b["ms_per_step_per_rank"]["max"] == b["ms_per_step_per_rank"]["min"] > 0
    assert b["ranks_seen_in_reduce"] == 1 and b["parity_per_rank"] == [True]


def test_bench_appends_configs_3_4_5_to_the_one_line():
    """The driver's N = 1 command measures the headline and then configs 3, 4 and 5 in the same process (here at small sizes):
    one GPU's share of the 8-GPU configs (c4_shard, c5_shard) and the whole configs (c4, c5: the N = 1 point of the strong-scaling
    curve); each carries value, roofline, cpu_baseline and the parity flag."""
    j = _bench("--rays", "32768", "--steps", "2", "--warmup", "1", "--extra-configs", "--extra-rays", "32768")
    assert set(j["configs"]) == {"c3", "c2_quads", "c4_shard", "c5_shard", "c4", "c5"}
    want = {"c3": OCTREE_KERNEL, "c2_quads": "hare_voxel_pool_quad", "c4_shard": "hare_voxel_pool_tri_g", "c5_shard": "hare_voxel_pool_tri_g",
            "c4": "hare_voxel_pool_tri_g", "c5": "hare_voxel_pool_tri_g"}
    for name, sub in j["configs"].items():
        assert sub["x_event_parity_vs_oracle"] is True, name
        assert sub["value"] > 0 and sub["roofline"]["frac"] > 0 and sub["roofline"]["kernel"] == want[name]
        assert sub["cpu_baseline"]["kind"] == "port" and sub["cpu_baseline"]["value"] > 0
        assert sub["cpu_baseline"]["host_cores"] >= sub["cpu_baseline"]["cores"] >= 1
        assert sub["scaling"] == ("strong" if name in ("c4", "c5") else "weak")
    assert j["configs"]["c5"]["roofline"]["live_casts_per_pass"] > 32768 * 7
    assert j["configs"]["c5_shard"]["bounce_batch"]["parity_vs_oracle"] is True
    for name in ("c2_quads", "c4_shard", "c5_shard", "c4", "c5"):        # the pool kernel has a counting build: the kernel's OWN bytes, which cannot pass the peak
        own = j["configs"][name]["roofline"]["own"]
        assert 0 < own["frac"] <= 1.0 and own["per_cast"]["T"] > 0 and own["per_cast"]["C"] > 0, (name, own)
    assert j["configs"]["c3"]["roofline"]["own"].get("frac") is None      # 32 768 rays go to hare_octree_group, which has none: the line says why
    assert "counting build" in j["configs"]["c3"]["roofline"]["own"]["why"]
    assert j["metric"].startswith("Mrays/s") and j["x_event_parity_vs_oracle"] is True     # the headline fields are untouched


@pytest.mark.parametrize("extra,kernel", [
    ((), "hare_voxel_pool_tri"),                          # the pool kernel serves every batch size (launch.cpp: choose_kernel)
    (("--kind", "octree"), OCTREE_KERNEL),
    (("--bounces", "3"), "hare_voxel_pool_tri"),
    (("--rays", "1048576"), "hare_voxel_pool_tri"),
    (("--kind", "kdtree", "--scene", "shoebox"), "hare_kdtree_dense"),      # K3d: the kd-tree's production kernel, with its counting build
    (("--scene", "hall_quads"), "hare_voxel_pool_quad"),                     # the quadrilateral build of the pool kernel
])
def test_bench_single_gpu_lines_carry_roofline_and_cpu_baseline(extra, kernel):
    j = _bench(*(("--rays", "32768") if "--rays" not in extra else ()), "--steps", "2", "--warmup", "1", *extra)
    assert j["n_gpus"] == 1 and j["x_event_parity_vs_oracle"] is True
    rf, cpu = j["roofline"], j["cpu_baseline"]
    assert rf["kernel"] == kernel and rf["bound"] == "hbm" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["algorithmic_bytes_per_launch"] > 104 * j["config"]["rays_per_gpu"] * 0.5
    assert cpu["kind"] == "port" and cpu["value"] > 0 and cpu["cores"] >= 1
    assert "timed region" in rf["kernel_ms_source"] or "event pair" in rf["kernel_ms_source"]     # ONE estimator, named
    if "--bounces" not in extra:          # what a caller that alternates two streams gets: beside the contract's value, never instead of it
        assert j["two_streams"]["value"] > 0.5 * j["value"] and j["two_streams"]["steps"] >= 2
    if "octree" not in extra:             # (32 768 octree rays go to hare_octree_group, which has no counting build)
        own = rf["own"]          # counted by the counting build of the kernel that was timed, on the same rays
        assert 0 < own["frac"] <= 1.0 and own["bytes_per_launch"] >= 104 * j["config"]["rays_per_gpu"]
        assert own["per_cast"]["L"] >= own["per_cast"]["K"] >= own["per_cast"]["T"] > 0


def test_bench_inproc_sharded_leg_with_two_scenes_on_this_box_one_gpu():
    """the path a one-process caller takes on a multi-GPU node -- ONE hare_shoot_batch_sharded and ONE
    hare_bounce_batch_sharded call over one scene per device -- is in the line as `inproc_sharded` at every N > 1.  Here at N = 1 with two
    scenes on this box's one device (the placement rule is scene k -> device k % device_count): rate, per-device kernel names, and
    parity of the concatenated events (the shoot's, and the last cast of the 8-bounce loop) against the oracle on a sample that touches
    both shards."""
    n = 131072
    j = _bench("--rays", str(n), "--steps", "2", "--warmup", "1", "--no-extra-configs", "--no-e2e", "--inproc-scenes", "2")
    s = j["inproc_sharded"]
    assert "error" not in s, s
    assert s["scenes"] == 2 and s["devices"] == [0, 0] and s["rays_total"] == n
    assert s["kernels"] == ["hare_voxel_pool_tri"]
    assert s["shoot"]["rays"] == n and s["shoot"]["hits"] == j["hits"] and s["shoot"]["mrays_s"] > 0
    assert n < s["bounce"]["casts"] <= 8 * n and s["bounce"]["mcasts_s"] > 0
    assert s["parity_vs_oracle"] is True
    k = _bench("--rays", str(n), "--steps", "2", "--warmup", "1", "--no-extra-configs", "--no-e2e")
    assert "inproc_sharded" not in k                       # off by default at N = 1
