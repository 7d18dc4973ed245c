"""Round-3 GPU tests: the launch-slot ring (no memset / reduce kernel around a launch; more than 64 launches in flight),
buffer-overlap rejection, per-scene options."""
import ctypes as C

import numpy as np
import pytest

import hare_amd as H
from hare_amd import capi
from oracle import pyoracle as po
from tests.helpers import assert_events_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hall():
    m = H.scenes.hall()
    return m, H.Topology(m.verts, m.nverts), po.Topology(m.verts, m.nverts)


@pytest.mark.parametrize("kind", ["voxel_persist", "voxel_pool", "octree_persist", "octree_pool"])
def test_more_launches_in_flight_than_launch_slots(hall, kind):
    """A scene keeps the scratch of its persistent launches (ticket word, done counters, counter shards) in a ring of 64
    slots which the launches themselves leave zeroed.  200 launches are queued here without a host synchronisation, round
    robin over five streams: the 65th must be ordered behind the first (it shares its slot), every launch must find its slot
    zeroed, and the counters the launches' last waves add up must be exact."""
    import torch
    m, T, To = hall
    if kind.startswith("voxel"):
        g, o = H.Voxel_Grid([T], 64), po.VoxelGrid([To], domain=64)
        g.set_option("voxel_kernel", 2 if kind.endswith("pool") else 1)
    else:
        g, o = H.Octree([T], 8, 16), po.Octree([To], 8, 16)
        g.set_option("octree_kernel", 2 if kind.endswith("pool") else 1)
    n = 20_000
    L = 200 if kind.startswith("voxel") else 80
    rays = H.scenes.burst_rays(n, m.size)
    ref, rc = o.shoot(rays, nthreads=8)
    streams = [torch.cuda.Stream() for _ in range(5)]
    d_rays = torch.from_numpy(rays).cuda()
    outs = [torch.zeros(n * 56, dtype=torch.uint8, device="cuda") for _ in range(L)]
    ctrs = torch.zeros((L, 8), dtype=torch.int64, device="cuda")
    total = torch.zeros(8, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    for k in range(L):
        st = streams[k % len(streams)]
        g.shoot_device(n, d_rays.data_ptr(), outs[k].data_ptr(), d_counters=ctrs[k].data_ptr(), stream=st.cuda_stream)
        g.shoot_device(n, d_rays.data_ptr(), outs[k].data_ptr(), d_counters=total.data_ptr(), stream=st.cuda_stream)
    torch.cuda.synchronize()
    c = ctrs.cpu().numpy()
    assert (c[:, 0] == n).all() and (c[:, 1] == rc["hits"]).all()
    assert (int(total[0]), int(total[1])) == (L * n, L * rc["hits"])          # counters ACCUMULATE, also across streams
    for k in (0, 1, 63, 64, 65, L - 1):
        got = np.frombuffer(outs[k].cpu().numpy().tobytes(), dtype=capi.XEVENT_DTYPE)
        assert_events_equal(got, ref, what=f"{kind}, launch {k}")


def test_overlapping_device_buffers_are_rejected(hall):
    """A live ray's own X_Event slot is its scratch in the pool kernels and rays are re-read while events are written: a call
    whose buffers alias would give wrong results silently -- it is refused instead."""
    import torch
    m, T, _ = hall
    g = H.Voxel_Grid([T], 16)
    n = 1000
    buf = torch.zeros(n * 56 + n * 48, dtype=torch.uint8, device="cuda")
    ctr = torch.zeros(8, dtype=torch.int64, device="cuda")
    base = buf.data_ptr()
    with pytest.raises(H.HareError) as e:
        g.shoot_device(n, base, base + 8)                            # events on top of the rays
    assert e.value.code == capi.HARE_E_INVALID and "overlap" in str(e.value)
    with pytest.raises(H.HareError):
        g.shoot_device(n, base, base + n * 48 - 8)                   # tail of the rays under the first event
    with pytest.raises(H.HareError):
        g.shoot_device(n, base, base + n * 48, d_counters=base + n * 48 + 56)    # counters inside the events
    g.shoot_device(n, base, base + n * 48, d_counters=ctr.data_ptr())            # adjacent is fine
    torch.cuda.synchronize()
    assert int(ctr[0]) == n


def test_environment_overrides_need_the_opt_in(hall, monkeypatch):
    """HARE_VOXEL_KERNEL & co. are read once, when a scene is created, and only in a process that set HARE_DEV=1: a stray
    variable in a production environment changes nothing."""
    m, T, _ = hall
    monkeypatch.delenv("HARE_DEV", raising=False)
    monkeypatch.setenv("HARE_VOXEL_KERNEL", "pool")
    g = H.Voxel_Grid([T], 64)
    assert g.kernel_name(1000) == "hare_voxel_persist_tri"
    monkeypatch.setenv("HARE_DEV", "1")
    assert g.kernel_name(1000) == "hare_voxel_persist_tri"           # not re-read by an existing scene
    g2 = H.Voxel_Grid([T], 64)
    assert g2.kernel_name(1000) == "hare_voxel_pool_tri"
    monkeypatch.setenv("HARE_VOXEL_KERNEL", "persist")
    assert g2.kernel_name(1000) == "hare_voxel_pool_tri"
