"""hare_topology_ingest (hare_amd/csrc/ingest.cpp) against the oracle's restatement of Topology(Point[][])
(Hare_Geometry_Topology.cs:120-142, :258-311, :342-377; Hash2 Hare_Geometry_Primitives.cs:237-250) and
against closed-form answers.  CPU only: this is host logic in front of the ray path."""
import numpy as np
import pytest

import hare_amd as H
from hare_amd import capi
from oracle import pyoracle as po
from tests.helpers import soup


def jittered_soup(seed):
    """A soup whose polygons share corners up to sub-millimetre noise, plus digits beyond 1e-15."""
    rng = np.random.default_rng(seed)
    v, nv, _ = soup(n_tri=300, n_quad=80, seed=seed)
    v = v.copy()
    # make many corners near-duplicates of other polygons' corners
    P = v.shape[0]
    for _ in range(400):
        a, b = rng.integers(0, P, 2)
        ca, cb = rng.integers(0, nv[a]), rng.integers(0, nv[b])
        v[b, cb] = v[a, ca] + rng.uniform(-4e-4, 4e-4, 3)
    v += rng.uniform(-1e-17, 1e-17, v.shape)          # below the Round(15) quantum
    for p in range(P):
        v[p, nv[p]:] = 0.0
    return np.ascontiguousarray(v), nv


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_ingest_matches_oracle_bit_for_bit(seed):
    v, nv = jittered_soup(seed)
    top = H.Topology.from_polygons(v, nv)
    ref = po.Topology(v, nv, ingest=True)
    assert np.array_equal(top.verts, ref.verts)
    assert top.vertices.shape[0] == ref.vertex_count
    assert top.vertices.shape[0] < int(nv.sum())       # something was merged
    assert np.array_equal(top.normals, ref.normals)
    assert np.array_equal(top.Min, ref.min) and np.array_equal(top.Max, ref.max)
    # corner_vertex indexes Vertices_List and reproduces verts
    for p in range(0, v.shape[0], 7):
        for c in range(nv[p]):
            assert np.array_equal(top.vertices[top.corner_vertex[p, c]], top.verts[p, c])
        assert (top.corner_vertex[p, nv[p]:] == -1).all()


def test_ingest_known_answers():
    # two triangles sharing an edge; the second one's shared corners are off by 0.2 mm / 0.3 mm but stay in
    # the same Hash2 millimetre cell (offsets from Modspace.Min = -1e-12): they snap onto the first's corners
    t0 = [(0.0, 0.0, 0.0), (1.0005, 0.0, 0.0), (0.0, 2.0005, 0.0)]
    t1 = [(1.0007, 0.0, 0.0), (0.0, 2.0008, 0.0), (1.5, 2.5, 0.25)]
    top = H.Topology.from_polygons([t0, t1])
    assert top.vertices.shape[0] == 4
    assert top.corner_vertex[:, :3].tolist() == [[0, 1, 2], [1, 2, 3]]
    assert top.verts[1, 0].tolist() == [1.0005, 0.0, 0.0] and top.verts[1, 1].tolist() == [0.0, 2.0005, 0.0]
    # a corner one millimetre cell further is NOT merged
    t2 = [(1.0015, 0.0, 0.0), (0.0, 2.0005, 0.0), (1.5, 2.5, 0.25)]
    top2 = H.Topology.from_polygons([t0, t2])
    assert top2.vertices.shape[0] == 5
    # Math.Round(x, 15): digits beyond 1e-15 are dropped, ties to even
    x = 0.1234567890123456789
    top3 = H.Topology.from_polygons([[(x, 0.0, 0.0), (1.0, 0.0, 0.0), (0.0, 1.0, 0.0)]])
    assert top3.verts[0, 0, 0] == po.lib().ho_dotnet_round(x, 15) == round(x * 1e15) / 1e15


def test_ingest_identity_on_lattice_scenes():
    # the synthetic scenes are snapped to 2^-8 m with distinct corners >= 1 mm apart: ingest changes nothing
    m = H.scenes.shoebox()
    top = H.Topology.from_polygons(m.verts, m.nverts)
    assert np.array_equal(top.verts, m.verts)
    g = H.Voxel_Grid([top], 8)
    g0 = H.Voxel_Grid([H.Topology(m.verts, m.nverts)], 8)
    assert np.array_equal(g.Voxel_Inv()[1], g0.Voxel_Inv()[1])


def test_ingest_errors():
    with pytest.raises(NotImplementedError):
        H.Topology.from_polygons([[(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0.5, 2, 0)]])
    v = np.zeros((1, 4, 3)); nv = np.array([5], np.int32); out = np.zeros((1, 4, 3))
    assert capi.lib.hare_topology_ingest(capi.ptr(v), capi.ptr(nv), 1, capi.ptr(out), None, None, None) == capi.HARE_E_UNSUPPORTED
    assert "3 or 4 sides" in capi.last_error()
    assert capi.lib.hare_topology_ingest(None, None, 1, None, None, None, None) == capi.HARE_E_INVALID
    assert capi.lib.hare_topology_ingest(None, None, 0, None, None, None, None) == capi.HARE_OK      # empty soup
