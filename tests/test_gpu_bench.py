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
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def _oracle_hits(n_total, scene="hall", domain=64):
    from oracle import pyoracle as po
    mesh = H.scenes.SCENES[scene]()
    rays = H.scenes.burst_rays(n_total, mesh.size)
    ev, _ = po.VoxelGrid([po.Topology(mesh.verts, mesh.nverts)], domain=domain).shoot(rays, nthreads=8)
    return int(ev["hit"].sum())


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


def test_bench_appends_configs_3_4_5_to_the_one_line():
    """The driver's N = 1 command measures the headline and then configs 3, 4 (shard) and 5 in the same process (here at
    small sizes): each carries value, roofline, cpu_baseline and the parity flag."""
    j = _bench("--rays", "32768", "--steps", "2", "--warmup", "1", "--extra-configs", "--extra-rays", "32768")
    assert set(j["configs"]) == {"c3", "c4_shard", "c5"}
    want = {"c3": "hare_octree_persist", "c4_shard": "hare_voxel_pool_tri_g", "c5": "hare_voxel_pool_tri_g"}
    for name, sub in j["configs"].items():
        assert sub["x_event_parity_vs_oracle"] is True, name
        assert sub["value"] > 0 and sub["roofline"]["frac"] > 0 and sub["roofline"]["kernel"] == want[name]
        assert sub["cpu_baseline"]["kind"] == "port" and sub["cpu_baseline"]["value"] > 0
    assert j["configs"]["c5"]["roofline"]["live_casts_per_pass"] > 32768 * 7
    assert j["metric"].startswith("Mrays/s") and j["x_event_parity_vs_oracle"] is True     # the headline fields are untouched


@pytest.mark.parametrize("extra,kernel", [
    ((), "hare_voxel_pool_tri"),                          # the pool kernel serves every batch size (api.cpp: choose_kernel)
    (("--kind", "octree"), "hare_octree_persist"),
    (("--bounces", "3"), "hare_voxel_pool_tri"),
    (("--rays", "1048576"), "hare_voxel_pool_tri"),
])
def test_bench_single_gpu_lines_carry_roofline_and_cpu_baseline(extra, kernel):
    j = _bench(*(("--rays", "32768") if "--rays" not in extra else ()), "--steps", "2", "--warmup", "1", *extra)
    assert j["n_gpus"] == 1 and j["x_event_parity_vs_oracle"] is True
    rf, cpu = j["roofline"], j["cpu_baseline"]
    assert rf["kernel"] == kernel and rf["bound"] == "hbm" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["algorithmic_bytes_per_launch"] > 104 * j["config"]["rays_per_gpu"] * 0.5
    assert cpu["kind"] == "port" and cpu["value"] > 0 and cpu["cores"] >= 1
