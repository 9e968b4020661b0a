"""Row G: graph tables vs. the reference's own output (tests/golden/graph_tables.json) and the
known answers listed in SURVEY.md 8(a)."""
import hashlib
import json
import os

import numpy as np
import pytest

from kinetic_gan_amd.graph import Graph_h36m, build_graph, graph_ntu


@pytest.fixture(scope="module")
def gold(golden_dir):
    return json.load(open(os.path.join(golden_dir, "graph_tables.json")))


@pytest.mark.parametrize("name,cls", [("ntu", graph_ntu), ("h36m", Graph_h36m)])
def test_tables_match_reference(gold, name, cls):
    g, ref = cls(), gold[name]
    assert [int(v) for v in g.num_node] == ref["num_node"]
    assert [int(v) for v in g.center] == ref["center"]
    for a, b in zip(g.map, ref["map"]):
        assert np.array_equal(np.asarray(a), np.asarray(b))
    for a, b in zip(g.edge, ref["edge"]):
        assert np.array_equal(np.asarray(a), np.asarray(b))
    assert len(g.mapping) == len(ref["mapping"])
    for la, lb in zip(g.mapping, ref["mapping"]):
        assert [np.asarray(h).tolist() for h in la] == lb
    for a, b in zip(g.As, ref["As"]):
        assert np.array_equal(a, np.asarray(b))      # bit-exact float64


def test_known_answers_ntu():
    g = graph_ntu()
    assert g.num_node == [25, 11, 5, 1] and g.center == [20, 10, 4, 0]
    assert g.keep(0).tolist() == [0, 2, 5, 7, 9, 11, 13, 14, 17, 18, 20]
    assert g.keep(1).tolist() == [2, 4, 6, 8, 10] and g.keep(2).tolist() == [4]
    nnz = [[int((a[k] != 0).sum()) for k in range(3)] for a in g.As]
    assert nnz == [[25, 40, 8], [11, 12, 8], [5, 6, 4], [1, 0, 0]]
    sha = [hashlib.sha256(np.ascontiguousarray(a, dtype=np.float32).tobytes()).hexdigest()[:16] for a in g.As]
    assert sha == ["0731c361914a9376", "26c22db2c71496a4", "76380b47f687b26e", "480376c6bf738a02"]
    for a in g.As:
        np.testing.assert_allclose(a.sum(0).sum(0), 1.0, atol=1e-12)   # every column of sum_k A_k sums to 1


def test_known_answers_h36m():
    g = Graph_h36m()
    assert g.num_node == [16, 7, 2, 1] and g.center == [8, 3, 1, 0]
    assert g.keep(0).tolist() == [0, 2, 5, 8, 9, 11, 14] and g.keep(1).tolist() == [0, 3] and g.keep(2).tolist() == [1]
    nnz = [[int((a[k] != 0).sum()) for k in range(3)] for a in g.As]
    assert nnz[:3] == [[16, 23, 7], [7, 6, 6], [2, 1, 1]]
    sha = [hashlib.sha256(np.ascontiguousarray(a, dtype=np.float32).tobytes()).hexdigest()[:16] for a in g.As]
    assert sha == ["e8b872de02f7fffc", "8ebe9111dc44957d", "c17c0f8b16d1dd74", "480376c6bf738a02"]


@pytest.mark.parametrize("ds", ["ntu", "h36m"])
def test_upsample_matrix_column_sums(ds):
    g = build_graph(ds)
    for lvl in (0, 1):
        np.testing.assert_allclose(g.upsample_matrix(lvl).sum(0), 1.0)
    s = g.upsample_matrix(2).sum(0)
    assert s.tolist() == ([0.5, 0.5, 0.5, 0.5, 1.0] if ds == "ntu" else [0.5, 1.0])   # the /2 quirk, generator.py:195
