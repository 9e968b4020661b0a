"""ORACLE support (test infrastructure, not product code): the reference's skeleton-graph tables as DATA.

``tests/golden/graph_tables.json`` holds what the reference's own ``graph_ntu()`` / ``Graph_h36m()``
(models/init_gan/graph_ntu.py:5-144, graph_h36m.py:5-204) produced when tests/golden/make_fixtures.py imported
them in the dev container: ``As`` (float64), ``map``, ``mapping``, ``num_node``, ``center``, ``edge``.  The oracle's
modules read their adjacencies / kept-vertex lists / up-sampling neighbourhoods from THAT file, not from the
product's ``kinetic_gan_amd.graph`` - the product's table builder is a thing under test
(tests/test_graph_tables.py), not part of the definition of the expected result.
"""
from __future__ import annotations

import json
import os
from types import SimpleNamespace

import numpy as np

_JSON = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "graph_tables.json")
_cache = {}


def load_graph(dataset: str) -> SimpleNamespace:
    """Object with the attributes the reference's graph classes expose to the models: ``As`` (list of (3, V, V)
    float64 arrays, graph_ntu.py:17-21), ``map`` (list of (V_l, 2) int arrays, :60-66), ``mapping`` (list of lists of
    int arrays ``[new_fine_idx, coarse_nbr...]``, :100-126), ``num_node``, ``center``."""
    name = "h36m" if dataset == "h36m" else "ntu"
    g = _cache.get(name)
    if g is None:
        with open(_JSON) as f:
            t = json.load(f)[name]
        g = SimpleNamespace(
            As=[np.asarray(a, dtype=np.float64) for a in t["As"]],
            map=[np.asarray(m, dtype=np.int64) for m in t["map"]],
            mapping=[[np.asarray(h, dtype=np.int64) for h in lvl] for lvl in t["mapping"]],
            num_node=[int(v) for v in t["num_node"]],
            center=[int(v) for v in t["center"]],
            edge=[np.asarray(e, dtype=np.int64) for e in t["edge"]],
        )
        _cache[name] = g
    return g
