"""N > 1 host logic on CPU (gloo, world_size 2): contiguous ray shards tile the batch, per-shard
outputs concatenate to exactly the single-process output, and the all-reduced hit counter equals
the whole-batch count.  The oracle stands in for the shooter here (checker role only): what is
under test is the sharding / reduction logic bench.py uses, not the kernel."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import hare_amd as H
from hare_amd.sharding import reduce_counters, shard_range


def test_shard_ranges_tile_exactly():
    for n in (0, 1, 7, 1000, 1 << 20, (1 << 24) + 5):
        for w in (1, 2, 3, 4, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_burst_shards_equal_slices_of_the_whole_burst():
    size = (40.0, 25.0, 18.0)
    whole = H.scenes.burst_rays(10007, size)
    for w in (2, 4, 8):
        parts = [H.scenes.burst_rays(10007, size, start=a, count=b - a) for a, b in (shard_range(10007, k, w) for k in range(w))]
        assert np.array_equal(np.concatenate(parts), whole)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as po
    mesh = H.scenes.shoebox()
    lo, hi = shard_range(n_total, rank, world)
    rays = H.scenes.burst_rays(n_total, mesh.size, start=lo, count=hi - lo)
    ev, ctr = po.VoxelGrid([po.Topology(mesh.verts, mesh.nverts)], domain=8).shoot(rays)
    c = torch.tensor([ctr["rays"], ctr["hits"], 0, 0, 0, 0, 0, 0], dtype=torch.int64)
    reduce_counters(c, dist)
    np.save(os.path.join(tmp, f"ev{rank}.npy"), ev)
    if rank == 0:
        np.save(os.path.join(tmp, "ctr.npy"), c.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_concatenate_to_single_process_output(tmp_path):
    from oracle import pyoracle as po
    n_total = 5001
    mp.spawn(_worker, args=(2, _free_port(), n_total, str(tmp_path)), nprocs=2, join=True)
    mesh = H.scenes.shoebox()
    rays = H.scenes.burst_rays(n_total, mesh.size)
    whole, ctr = po.VoxelGrid([po.Topology(mesh.verts, mesh.nverts)], domain=8).shoot(rays)
    got = np.concatenate([np.load(tmp_path / "ev0.npy"), np.load(tmp_path / "ev1.npy")])
    assert got.tobytes() == whole.tobytes()
    c = np.load(tmp_path / "ctr.npy")
    assert c[0] == n_total and c[1] == ctr["hits"] == int(whole["hit"].sum())
