"""The mutex-watershed restatement (oracle/mws_ref.py, parity unpinned) on hand-built cases whose answer follows from
the published rule alone, and the host entry point of the library against it (no GPU involved: bsmi_mws_cluster is
host code)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import mws_ref  # noqa: E402


def test_chain_with_a_long_range_mutex():
    # 0 -0.9- 1 -0.8- 2 -0.7- 3, and a repulsive 0 ~ 2 of 0.85: {0,1} is formed first, then 0 ~ 2 forbids joining 2
    edges = [(0, 1), (1, 2), (2, 3), (0, 2)]
    scores = [0.9, 0.8, 0.7, -0.85]
    assert mws_ref.mws_cluster(4, edges, scores).tolist() == [1, 1, 3, 3]
    # weaker repulsion: the attractive chain wins before the constraint is seen, which then changes nothing
    assert mws_ref.mws_cluster(4, edges, [0.9, 0.8, 0.7, -0.5]).tolist() == [1, 1, 1, 1]


def test_constraints_are_inherited_by_unions():
    # a ~ c is forbidden; a joins b; then b - c (attractive, weaker) must be refused because b's cluster holds a
    edges = [(0, 2), (0, 1), (1, 2)]
    assert mws_ref.mws_cluster(3, edges, [-0.9, 0.8, 0.7]).tolist() == [1, 1, 3]
    # zero and NaN weights are no edges
    assert mws_ref.mws_cluster(3, edges, [0.0, float("nan"), 0.7]).tolist() == [1, 2, 2]


def test_grid_edges_and_strides():
    a = np.zeros((2, 1, 2, 4))
    a[0] = 0.5    # offset (0, 0, -1): attractive along x
    a[1] = -0.9   # offset (0, -1, 0): repulsive between the two rows
    lab = mws_ref.mws_agglom(a, [[0, 0, -1], [0, -1, 0]])
    assert lab.tolist() == [[[1, 1, 1, 1], [5, 5, 5, 5]]]
    # stride 2 along x on the attractive channel: only the edges leaving x = 2 survive (x = 0 has no left neighbour)
    lab = mws_ref.mws_agglom(a, [[0, 0, -1], [0, -1, 0]], strides=[[1, 1, 2], [1, 1, 1]])
    assert lab.tolist() == [[[1, 2, 2, 4], [5, 6, 6, 8]]]


def test_library_graph_clustering_equals_restatement():
    from bootstrapper_amd.post.mws import mws_cluster
    rng = np.random.default_rng(0)
    for n, m in [(1, 0), (2, 1), (30, 200), (200, 3000)]:
        e = rng.integers(0, n, size=(m, 2))
        e = e[e[:, 0] != e[:, 1]] if m else e
        s = np.round(rng.standard_normal(len(e)), 1)
        assert np.array_equal(mws_cluster(n, e, s), mws_ref.mws_cluster(n, e, s))
