"""A loose octree whose nodes shrink below the reference's absolute 0.1 m padding grows 8x per level ("Octree - alt.cs":99-111,
DESIGN.md F16): a careless maxDepth asks for more nodes than any machine holds.  Round 2 recorded a segmentation fault there
(an unchecked realloc in the oracle, rc 139).  Both the oracle and the product now stop at a budget with a clean error."""
import numpy as np
import pytest

import hare_amd as H
from hare_amd import capi
from oracle import pyoracle as po
from tests.helpers import soup


def test_exploding_octree_is_a_clean_error_in_oracle_and_product(monkeypatch):
    monkeypatch.setenv("HARE_BUILD", "host")          # the host builder (this test must not need a GPU; the GPU builder shares the budget)
    v, nv, _ = soup(n_tri=60, n_quad=0, seed=1)       # 6 x 5 x 4 m: nodes of 0.4 m at level ~6, so 17 levels cannot be finite
    with pytest.raises(MemoryError) as e:
        po.Octree([po.Topology(v, nv)], 17, 1)
    assert "2^24 nodes" in str(e.value) or "2^28" in str(e.value)
    with pytest.raises(H.HareError) as e:
        H.Octree([H.Topology(v, nv)], 17, 1)
    assert e.value.code == capi.HARE_E_NOMEM and "budget" in str(e.value)
    # and the library is fine afterwards: the same soup at a depth its extent supports
    a = H.Octree([H.Topology(v, nv)], 4, 1)
    b = po.Octree([po.Topology(v, nv)], 4, 1)
    assert a.info().n_nodes == b.n_nodes


def test_oracle_builders_report_failure_instead_of_crashing():
    """ho_*_build return NULL with a message on allocation failure / budget; pyoracle turns that into MemoryError."""
    L = po.lib()
    assert L.ho_last_error() is not None
    v, nv, _ = soup(n_tri=40, n_quad=10, seed=2)
    T = po.Topology(v, nv)
    with pytest.raises(MemoryError):
        po.KDTree([T], 40, 0)                          # every polygon straddles some split: lists double per level
    assert po.KDTree([T], 6, 4).n_nodes > 1


@pytest.mark.parametrize("kind", ["kdtree", "octree"])
@pytest.mark.parametrize("where", ["first allocation", "stack growth"])
def test_a_failed_traversal_allocation_raises_and_is_never_a_miss_record(kind, where):
    """oracle/ho_kdtree.c and ho_octree.c: a ray whose traversal stack cannot be allocated or grown is an ERROR (-1), and
    pyoracle.shoot raises -- a checker that wrote a miss record there could pass an allocation failure off as a result."""
    L = po.lib()
    v, nv, _ = soup(n_tri=80, n_quad=0, seed=3)
    T = po.Topology(v, nv)
    tree = po.KDTree([T], 8, 2) if kind == "kdtree" else po.Octree([T], 4, 2)
    rays = H.scenes.random_rays(64, (6.0, 5.0, 4.0))
    ref, _ = tree.shoot(rays)
    assert ref["hit"].sum() > 0
    try:
        if where == "stack growth":
            L.ho_test_stack_cap(2 if kind == "kdtree" else 1)    # the stack must grow on the first interior node
            grown, _ = tree.shoot(rays)                          # ... and a stack that CAN grow changes nothing
            for f in ref.dtype.names:
                assert np.array_equal(grown[f], ref[f])
            L.ho_test_fail_alloc_after(1)                        # the stack itself is allocated; its first growth fails
        else:
            L.ho_test_fail_alloc_after(0)
        with pytest.raises(MemoryError):
            tree.shoot(rays)
    finally:
        L.ho_test_fail_alloc_after(-1)
        L.ho_test_stack_cap(0)
    again, _ = tree.shoot(rays)
    for f in ref.dtype.names:
        assert np.array_equal(again[f], ref[f])
