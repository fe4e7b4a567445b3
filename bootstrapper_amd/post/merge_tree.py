"""Merge tree over an agglomeration merge history.

Behavioural mirror of /root/reference/bootstrapper/post/merge_tree.py:30-113 (`MergeTree`):
replay `merge(u, v, target, score)` rows, then `find_merges(us, vs)` gives, per pair of
fragments, the score at which they first belong to the same segment (NaN if never or if a
fragment is unknown).  Pinned by tests/golden/host_cases.json (reference run).
"""
import numpy as np


class MergeTree:
    def __init__(self, leaf_nodes=None):
        self._index = {}     # tree-node id -> row
        self._level = []
        self._parent = []    # row of the parent merge node, -1 for roots
        self._score = []
        self.id_to_node = {}  # fragment / segment id -> its current top merge node id
        self.next_id = 0
        self.max_level = 0
        if leaf_nodes is not None:
            leaves = [int(n) for n in leaf_nodes]
            for n in leaves:
                if n not in self._index:
                    self._new(n, 0, 0.0)
                    self.id_to_node[n] = n
            self.next_id = max(leaves) + 1

    def _new(self, node_id, level, score):
        self._index[node_id] = len(self._level)
        self._level.append(level)
        self._parent.append(-1)
        self._score.append(score)
        return self._index[node_id]

    def merge(self, u, v, target, score):
        u, v, target = int(u), int(v), int(target)
        node = self.next_id
        self.next_id += 1
        ru = self._index[self.id_to_node[u]]
        rv = self._index[self.id_to_node[v]]
        level = max(self._level[ru], self._level[rv]) + 1
        self.max_level = max(self.max_level, level)
        r = self._new(node, level, float(score))
        self._parent[ru] = r
        self._parent[rv] = r
        self.id_to_node[target] = node

    def find_merges(self, us, vs):
        out = np.full(len(us), np.nan, dtype=np.float64)
        level, parent, score = self._level, self._parent, self._score
        for k, (u, v) in enumerate(zip(us, vs)):
            a = self._index.get(int(u), -1)
            b = self._index.get(int(v), -1)
            if a < 0 or b < 0:
                continue
            while a != b:
                if level[a] > level[b]:
                    a, b = b, a
                a = parent[a]  # climb from the lower node
                if a < 0:
                    break
            else:
                out[k] = score[a]
        return out

    def find_merge(self, u, v):
        r = self.find_merges((u,), (v,))[0]
        return None if np.isnan(r) else float(r)
