"""Octree / KDTree over SEVERAL topologies, as the reference has it -- not by design but as written, and the oracle restates it
as written: the constructors build a fresh root per topology and the LAST one's stays ("Octree - alt.cs":63-88, KDTree.cs:67-87);
polygon ids 0..P_last-1 are binned by the vertices (octree, :123) / centroids (kd, KDTree.cs:98-133) of Model[0]; the kd box grows
over all topologies; Shoot(top_index) intersects Model[top_index]'s polygons of those ids.  The product follows: host builders,
host single-ray path and the batch kernels, bit-identical to the oracle.  Where the reference would index out of range the
product returns an error instead."""
import numpy as np
import pytest

import hare_amd as H
from hare_amd import capi
from oracle import pyoracle as po
from tests.helpers import assert_events_equal, soup, soup_rays
from tests.test_shoot_one import shoot_all


def two_topologies():
    """Model[0]: 500 polygons; Model[1]: 380 polygons of another soup in a box shifted and shrunk (so that the root cube of the
    last topology differs from the first one's and some of Model[0]'s polygons fall outside it)."""
    v0, n0, size = soup()
    v1, n1, _ = soup(n_tri=300, n_quad=80, seed=9, size=(5.0, 4.5, 3.5))
    return (v0, n0), (v1, n1), size


@pytest.mark.parametrize("kind", ["octree", "kdtree"])
def test_trees_over_two_topologies_host_path_equals_oracle(kind, monkeypatch):
    monkeypatch.setenv("HARE_BUILD", "host")
    (v0, n0), (v1, n1), size = two_topologies()
    rays = soup_rays(1500 if kind == "octree" else 500, size)
    rng = np.random.default_rng(5)
    e1 = rng.integers(-1, len(n1), len(rays)).astype(np.int32)
    Ts, To = [H.Topology(v0, n0), H.Topology(v1, n1)], [po.Topology(v0, n0), po.Topology(v1, n1)]
    part, ora = (H.Octree(Ts, 5, 6), po.Octree(To, 5, 6)) if kind == "octree" else (H.KDTree(Ts, 6, 8), po.KDTree(To, 6, 8))
    for top in (0, 1):
        assert_events_equal(shoot_all(part, rays, top=top)[0], ora.shoot(rays, top_index=top)[0], what="%s top %d" % (kind, top))
        assert_events_equal(shoot_all(part, rays, e1=e1, top=top)[0], ora.shoot(rays, top_index=top, excl1=e1)[0],
                            what="%s top %d excl" % (kind, top))
    # the tree itself -- boxes, links, leaf lists in order -- is the oracle's (KDTree.Shoot visits every leaf, F4: its
    # events cannot tell one tree over the same ids from another, the structure can)
    for got, want in zip(part.nodes(), ora.export()):
        assert got.shape == want.shape and np.array_equal(got, want, equal_nan=True)
    # ... and it is not the tree a build over Model[1] alone gives (else this test would prove nothing)
    single = H.Octree([Ts[1]], 5, 6) if kind == "octree" else H.KDTree([Ts[1]], 6, 8)
    sn, pn = single.nodes(), part.nodes()
    assert sn[0].shape != pn[0].shape or not np.array_equal(sn[0], pn[0]), "two-topology tree has the one-topology tree's boxes"
    if kind == "octree":
        a, b = shoot_all(part, rays, top=1)[0], shoot_all(single, rays)[0]
        assert any(np.any(a[f] != b[f]) for f in ("hit", "poly_id", "t")), "two-topology octree behaves like the one-topology octree"


@pytest.mark.parametrize("kind", ["octree", "kdtree"])
def test_trees_refuse_what_the_reference_throws_on(kind, monkeypatch):
    monkeypatch.setenv("HARE_BUILD", "host")
    (v0, n0), (v1, n1), size = two_topologies()
    big_last = [H.Topology(v1, n1), H.Topology(v0, n0)]          # the last topology has MORE polygons than Model[0]
    make = (lambda Ts, d, p: H.Octree(Ts, d, p)) if kind == "octree" else (lambda Ts, d, p: H.KDTree(Ts, d, p))
    with pytest.raises(H.HareError, match="more polygons than topology 0"):
        make(big_last, 5, 6)
    # ... unless nothing is ever split (the reference then never looks at Model[0]): one leaf holding ids 0..499 --
    # shooting at topology 0 (380 polygons) would index it out of range, topology 1 is fine
    part = make(big_last, 0, 6)
    ora = (po.Octree if kind == "octree" else po.KDTree)([po.Topology(v1, n1), po.Topology(v0, n0)], 0, 6)
    rays = soup_rays(200, size)
    assert_events_equal(shoot_all(part, rays, top=1)[0], ora.shoot(rays, top_index=1)[0], what=kind + " unsplit, top 1")
    ev = np.zeros(1, capi.XEVENT_DTYPE)
    r = np.array(rays[0], np.float64)
    assert capi.lib.hare_shoot_one(part._h, part._kind, 0, r.ctypes.data, -1, -1, ev.ctypes.data) == capi.HARE_E_INVALID
    assert "does not have" in capi.last_error()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["octree", "kdtree"])
def test_trees_over_two_topologies_batch_kernels_equal_oracle(kind):
    (v0, n0), (v1, n1), size = two_topologies()
    rays = soup_rays(20000 if kind == "octree" else 3000, size)
    rng = np.random.default_rng(6)
    e1 = rng.integers(-1, len(n1), len(rays)).astype(np.int32)
    e2 = rng.integers(-1, len(n1), len(rays)).astype(np.int32)
    Ts, To = [H.Topology(v0, n0), H.Topology(v1, n1)], [po.Topology(v0, n0), po.Topology(v1, n1)]
    part, ora = (H.Octree(Ts, 5, 6), po.Octree(To, 5, 6)) if kind == "octree" else (H.KDTree(Ts, 6, 8), po.KDTree(To, 6, 8))
    for top in (0, 1):
        ev, ctr = part.Shoot_batch(rays, top_index=top)
        ref, rc = ora.shoot(rays, top_index=top, nthreads=8)
        assert_events_equal(ev, ref, what="%s batch top %d" % (kind, top))
        assert ctr["hits"] == rc["hits"]
        ev, _ = part.Shoot_batch(rays, top_index=top, poly_origin1=e1, poly_origin2=e2)
        assert_events_equal(ev, ora.shoot(rays, top_index=top, excl1=e1, excl2=e2, nthreads=8)[0], what="%s batch top %d excl" % (kind, top))
        one = shoot_all(part, rays[:300], top=top)[0]                       # host single-ray path == kernels
        assert_events_equal(one, ev_plain(part, rays[:300], top), what="%s one == batch" % kind)
    big_last = [H.Topology(v1, n1), H.Topology(v0, n0)]
    unsplit = (H.Octree if kind == "octree" else H.KDTree)(big_last, 0, 6)
    with pytest.raises(H.HareError, match="does not have"):
        unsplit.Shoot_batch(rays[:64], top_index=0)


def ev_plain(part, rays, top):
    return part.Shoot_batch(rays, top_index=top)[0]
