"""Skeleton graph tables for the st_gcn hot path (host side, numpy only).

Row G of SURVEY.md section 8(a).  Restates what the reference builds in
``models/init_gan/graph_ntu.py:5-208`` and ``models/init_gan/graph_h36m.py:5-204``:

* ``As[l]``      (K=3, V_l, V_l) float64 column-normalised adjacency split into the
                 partitions [self, root+closer, further]      (graph_ntu.py:117-144)
* ``map[l]``     (V_l, 2) int: [index at level l, label at level l-1] (graph_ntu.py:49,81-82)
* ``mapping[l]`` list of int arrays [new_fine_index, coarse_neighbour, ...] used by the
                 generator's spatial up-sampling                 (graph_ntu.py:184-208)
* ``num_node``, ``center``, ``edge``, ``nodes``, ``hop_dis``

The reference leans on networkx for the bookkeeping; this module does the same
coarsening heuristic on a tiny insertion-ordered adjacency structure of its own
(iteration order of nodes / incident edges is what the heuristic's result depends
on, so the ordering rules are kept explicit here).  The tables are additionally
exposed as dense matrices that the HIP path wants:

* ``upsample_matrix(l)``  U (V_{l+1}, V_l): the generator's ``upsample_s`` as a linear map
                          (generator.py:185-200, incl. the extra /2 when lvl == 2)
* ``keep(l)``             int32 vertex indices kept when going from level l to l+1
                          (discriminator.py:139-142)
"""
from __future__ import annotations

import numpy as np

__all__ = ["SkeletonGraph", "graph_ntu", "Graph_h36m", "build_graph"]


class _OrderedGraph:
    """Undirected simple graph; node and neighbour iteration follow insertion order."""

    def __init__(self, nodes=(), edges=()):
        self.adj: dict[int, dict[int, None]] = {}
        for n in nodes:
            self.add_node(n)
        for u, v in edges:
            self.add_edge(u, v)

    def add_node(self, n):
        self.adj.setdefault(int(n), {})

    def add_edge(self, u, v):
        u, v = int(u), int(v)
        self.add_node(u)
        self.add_node(v)
        self.adj[u][v] = None
        self.adj[v][u] = None

    def remove_node(self, n):
        for m in list(self.adj[n]):
            if m != n:
                del self.adj[m][n]
        del self.adj[n]

    def nodes(self):
        return list(self.adj)

    def __len__(self):
        return len(self.adj)

    def incident(self, n):
        """Edges (n, m) in the order the neighbours were attached."""
        return [(n, m) for m in self.adj[n]]

    def edge_list(self):
        """Each undirected edge once, reported from its earlier endpoint (node order)."""
        done, out = set(), []
        for u, nbrs in self.adj.items():
            for v in nbrs:
                if v not in done:
                    out.append((u, v))
            done.add(u)
        return out

    def relabelled(self, table):
        """New graph with labels table[n]; nodes first (in order), then edges in edge_list order."""
        g = _OrderedGraph(nodes=[table[n] for n in self.adj])
        for u, v in self.edge_list():
            g.add_edge(table[u], table[v])
        return g

    def compacted(self):
        return self.relabelled({n: i for i, n in enumerate(self.adj)})

    def first_basis_cycle(self):
        """First cycle of a DFS cycle basis, or None if the graph is a forest.

        Same traversal discipline as networkx.cycle_basis (pinned networkx==2.5 in the
        reference's requirements.txt:11; 3.4 in this image): roots are taken from the
        END of the node order, the frontier is a LIFO stack, and a non-tree edge closes a
        cycle through the predecessor chain.  Only the first cycle's length is consumed
        by the coarsening rule (graph_ntu.py:74-78).
        """
        pending = dict.fromkeys(self.adj)
        while pending:
            root = pending.popitem()[0]
            stack = [root]
            pred = {root: root}
            seen_from = {root: set()}
            while stack:
                z = stack.pop()
                zfrom = seen_from[z]
                for nb in self.adj[z]:
                    if nb not in seen_from:
                        pred[nb] = z
                        stack.append(nb)
                        seen_from[nb] = {z}
                    elif nb == z:
                        return [z]
                    elif nb not in zfrom:
                        stop = seen_from[nb]
                        cyc = [nb, z]
                        p = pred[z]
                        while p not in stop:
                            cyc.append(p)
                            p = pred[p]
                        cyc.append(p)
                        return cyc
            for n in pred:
                pending.pop(n, None)
        return None


def _hop_distance(num_node, edge, max_hop=1):
    # graph_ntu.py:148-160
    a = np.zeros((num_node, num_node))
    for i, j in edge:
        a[j, i] = 1
        a[i, j] = 1
    hop = np.full((num_node, num_node), np.inf)
    reach = [np.linalg.matrix_power(a, d) > 0 for d in range(max_hop + 1)]
    for d in range(max_hop, -1, -1):
        hop[reach[d]] = d
    return hop


def _column_normalise(a):
    # graph_ntu.py:163-171  (A . D^-1 with D = column sums)
    col = a.sum(0)
    scale = np.zeros_like(col)
    nz = col > 0
    scale[nz] = 1.0 / col[nz]
    return a * scale[None, :]


class SkeletonGraph:
    """4-level coarsened skeleton graph with partitioned adjacency per level."""

    lvls = 4
    # subclasses fill these
    bones: list = []
    joints: int = 0
    root_joint: int = 0

    def __init__(self, max_hop=1, dilation=1):
        self.max_hop = max_hop
        self.dilation = dilation
        self.As = []
        self.hop_dis = []
        self._coarsen()
        for lvl in range(self.lvls):
            self.hop_dis.append(_hop_distance(self.num_node[lvl], self.edge[lvl], max_hop))
            self.As.append(self._partitioned_adjacency(lvl))
        self.mapping = self._upsample_neighbourhoods()

    # hook for the one dataset-specific exception (graph_h36m.py:60)
    def _pinned(self, node, level):
        return False

    def _coarsen(self):
        # graph_ntu.py:26-114
        g = _OrderedGraph(nodes=range(self.joints), edges=self.bones).compacted()
        self.center = [self.root_joint]
        self.num_node, self.nodes, self.map, self.edge = [], [], [], []

        def record(graph, level_map):
            loops = [(n, n) for n in graph.nodes()]
            el = graph.edge_list()
            self.map.append(level_map)
            self.edge.append(np.array(el + loops) if el else loops)
            self.nodes.append(np.arange(len(graph)))
            self.num_node.append(len(graph))

        record(g, np.array([[i, n] for i, n in enumerate(g.nodes())]))

        for level in range(self.lvls - 1):
            stay = []
            degree = 1
            while True:
                drop = []
                for n in g.nodes():
                    if self._pinned(n, level):
                        continue
                    inc = g.incident(n)
                    if len(inc) == degree and n not in stay:
                        nbrs = [m for _, m in inc]
                        stay.extend(nbrs)
                        for a in nbrs:          # clique-connect the neighbours of a dropped node
                            for b in nbrs:
                                if a != b:
                                    g.add_edge(a, b)
                        drop.append(n)
                if degree > 10:
                    break
                for n in drop:
                    g.remove_node(n)
                cyc = g.first_basis_cycle()
                if cyc is not None and len(cyc) == len(g):
                    for n in [m for m in g.nodes() if m not in stay]:
                        g.remove_node(n)
                degree += 1

            survivors = g.nodes()
            level_map = np.array([[i, n] for i, n in enumerate(survivors)])
            table = {n: i for i, n in enumerate(survivors)}
            if self.center[-1] in table:
                self.center.append(table[self.center[-1]])
            g = g.relabelled(table).compacted()
            record(g, level_map)

        for name in ("num_node", "nodes", "edge", "center", "map"):
            assert len(getattr(self, name)) == self.lvls, name

    def _partitioned_adjacency(self, lvl):
        # graph_ntu.py:117-144
        v = self.num_node[lvl]
        hop = self.hop_dis[lvl]
        c = self.center[lvl]
        hops = range(0, self.max_hop + 1, self.dilation)
        adj = np.zeros((v, v))
        for h in hops:
            adj[hop == h] = 1
        norm = _column_normalise(adj)
        to_c = hop[:, c]
        # all matrices below are indexed [j, i] like the reference's double loop
        same = to_c[:, None] == to_c[None, :]      # d(j,c) == d(i,c)   (inf == inf counts as same)
        closer = to_c[:, None] > to_c[None, :]     # d(j,c) >  d(i,c)
        parts = []
        for h in hops:
            sel = hop == h
            root = np.where(sel & same, norm, 0.0)
            close = np.where(sel & ~same & closer, norm, 0.0)
            far = np.where(sel & ~same & ~closer, norm, 0.0)
            if h == 0:
                parts.append(root)
            else:
                parts.append(root + close)
                parts.append(far)
        return np.stack(parts)

    def _upsample_neighbourhoods(self):
        # graph_ntu.py:184-208 (returned list already reversed: index = level being up-sampled TO)
        out = []
        for fine in range(self.lvls - 1):
            coarse = fine + 1
            links = {(int(a), int(b)) for a, b in np.asarray(self.edge[fine]).tolist()}
            kept = self.map[coarse][:, 1]
            hoods = []
            for node in self.nodes[fine]:
                node = int(node)
                if node in kept:
                    continue
                hood = [int(ci) for ci, lab in self.map[coarse]
                        if (node, int(lab)) in links or (int(lab), node) in links]
                if hood:
                    hoods.append(np.array([node] + hood))
            out.append(hoods)
        return out

    # ---- dense forms used by the HIP path -------------------------------------------------
    def keep(self, lvl):
        """Vertex indices of level ``lvl`` that survive to level ``lvl+1`` (discriminator.py:140)."""
        return np.ascontiguousarray(self.map[lvl + 1][:, 1], dtype=np.int32)

    def upsample_matrix(self, lvl):
        """U with x_fine[..., w] = sum_v x_coarse[..., v] * U[v, w]  (generator.py:185-200).

        ``lvl`` is the level being up-sampled TO (the block's own ``lvl``).
        """
        vc = self.num_node[lvl + 1]
        cols = [("keep", c) for c in range(vc)]
        extra = 2.0 if lvl == 2 else 1.0
        for hood in self.mapping[lvl]:
            cols.insert(int(hood[0]), ("mean", [int(c) for c in hood[1:]]))
        u = np.zeros((vc, len(cols)))
        for w, (kind, arg) in enumerate(cols):
            if kind == "keep":
                u[arg, w] = 1.0
            else:
                for c in arg:
                    u[c, w] += 1.0 / (len(arg) * extra)
        assert u.shape[1] == self.num_node[lvl], (u.shape, self.num_node[lvl])
        return u


class graph_ntu(SkeletonGraph):
    """NTU RGB+D 25-joint skeleton, 25 -> 11 -> 5 -> 1 (graph_ntu.py:29-38)."""

    joints = 25
    root_joint = 21 - 1
    bones = [(i - 1, j - 1) for i, j in [
        (1, 2), (2, 21), (3, 21), (4, 3), (5, 21), (6, 5), (7, 6), (8, 7), (9, 21), (10, 9),
        (11, 10), (12, 11), (1, 13), (14, 13), (15, 14), (16, 15), (1, 17), (18, 17), (19, 18),
        (20, 19), (22, 8), (23, 8), (24, 12), (25, 12)]]


class Graph_h36m(SkeletonGraph):
    """Human3.6M 16-joint skeleton, 16 -> 7 -> 2 -> 1 (graph_h36m.py:29-39)."""

    joints = 16
    root_joint = 8
    bones = [(1, 2), (2, 3), (0, 1), (4, 5), (5, 6), (0, 4), (0, 7), (7, 8), (8, 9),
             (8, 10), (10, 11), (11, 12), (8, 13), (13, 14), (14, 15)]

    def _pinned(self, node, level):
        return node == 9 and level == 0   # graph_h36m.py:60


def build_graph(dataset="ntu"):
    """Same dispatch as generator.py:46 / discriminator.py:18."""
    return graph_ntu() if dataset == "ntu" else Graph_h36m()
