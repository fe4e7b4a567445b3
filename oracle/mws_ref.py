"""CPU restatement of the mutex watershed -- TEST INFRASTRUCTURE ONLY (imported by tests/ only).

PARITY UNPINNED.  The reference computes fragments with `mwatershed.agglom` (/root/reference/bootstrapper/post/mws.py:51-56),
fragment-pair affinities with volara `AffAgglom` and the global clustering with volara `GraphMWS`
(post/watershed_mutex.py:143-161).  mwatershed (Rust) and volara are third party and absent from /root/reference
and from this image, and the reference holds no fixtures for them, so this file restates the published algorithm
(Wolf et al., "The Mutex Watershed", ECCV 2018):

  * every offset k and voxel p with p + offset_k inside the volume give an edge of weight affs[k][p] (stride_k
    subsamples the voxels p: all coordinates divisible by the stride);
  * edges are visited by |weight| descending -- ties in (k, p) ascending order, a documented choice;
  * a positive edge unites the clusters of its ends unless a mutex constraint separates them, a negative one adds a
    constraint between them, a zero / NaN one does nothing;
  * a voxel's label is 1 + the smallest voxel index of its cluster (any labelling of the same partition is
    equivalent: results are compared after ID remap).

Deliberately naive (dict of sets, python loops): small cases only.
"""
import numpy as np


class _Forest:
    def __init__(self, n):
        self.parent = list(range(n))
        self.mutex = [set() for _ in range(n)]  # per root: roots it must stay apart from

    def find(self, x):
        while self.parent[x] != x:
            self.parent[x] = self.parent[self.parent[x]]
            x = self.parent[x]
        return x

    def edge(self, u, v, w):
        a, b = self.find(u), self.find(v)
        if a == b or not abs(w) > 0:
            return
        if b in self.mutex[a]:
            return
        if w < 0:
            self.mutex[a].add(b)
            self.mutex[b].add(a)
            return
        # unite: b into a; everything that excluded b now excludes a
        self.parent[b] = a
        for c in self.mutex[b]:
            self.mutex[c].discard(b)
            self.mutex[c].add(a)
            self.mutex[a].add(c)
        self.mutex[b] = set()

    def labels(self):
        n = len(self.parent)
        low = {}
        for i in range(n):
            low.setdefault(self.find(i), i)
        return np.array([low[self.find(i)] + 1 for i in range(n)], dtype=np.uint64)


def grid_edges(affs, offsets, strides=None):
    """[(k, p, q, w)] in (k, p) ascending order."""
    K, D, H, W = affs.shape
    out = []
    for k, (oz, oy, ox) in enumerate(offsets):
        st = strides[k] if strides is not None else (1, 1, 1)
        for z in range(D):
            for y in range(H):
                for x in range(W):
                    if not (0 <= z + oz < D and 0 <= y + oy < H and 0 <= x + ox < W):
                        continue
                    if z % st[0] or y % st[1] or x % st[2]:
                        continue
                    out.append((k, (z * H + y) * W + x, ((z + oz) * H + y + oy) * W + x + ox, float(affs[k, z, y, x])))
    return out


def mws_agglom(affs, offsets, strides=None):
    """labels u64 [D][H][W] of `mwatershed.agglom(affs, offsets, strides=strides)` as restated above."""
    affs = np.asarray(affs, dtype=np.float64)
    edges = [e for e in grid_edges(affs, offsets, strides) if abs(e[3]) > 0]
    order = sorted(range(len(edges)), key=lambda i: -abs(edges[i][3]))  # stable
    f = _Forest(int(np.prod(affs.shape[1:])))
    for i in order:
        _, p, q, w = edges[i]
        f.edge(p, q, w)
    return f.labels().reshape(affs.shape[1:])


def mws_cluster(n_nodes, edges, scores):
    order = sorted(range(len(edges)), key=lambda i: -abs(float(scores[i])))
    f = _Forest(int(n_nodes))
    for i in order:
        f.edge(int(edges[i][0]), int(edges[i][1]), float(scores[i]))
    return f.labels()


def pair_affinity(affs_u8, offsets, frags):
    """{(u, v): (sum of affinity bytes, count)} over voxel pairs (p, p + offset_k) whose fragments u < v differ and are non-zero."""
    K, D, H, W = affs_u8.shape
    out = {}
    for k, (oz, oy, ox) in enumerate(offsets):
        for z in range(D):
            for y in range(H):
                for x in range(W):
                    if not (0 <= z + oz < D and 0 <= y + oy < H and 0 <= x + ox < W):
                        continue
                    a, b = int(frags[z, y, x]), int(frags[z + oz, y + oy, x + ox])
                    if a == 0 or b == 0 or a == b:
                        continue
                    key = (min(a, b), max(a, b))
                    s, c = out.get(key, (0, 0))
                    out[key] = (s + int(affs_u8[k, z, y, x]), c + 1)
    return out


def same_partition(a, b):
    """True if the label arrays describe the same partition (bijection between their ids)."""
    a, b = np.asarray(a).ravel(), np.asarray(b).ravel()
    fwd, bwd = {}, {}
    for x, y in zip(a.tolist(), b.tolist()):
        if fwd.setdefault(x, y) != y or bwd.setdefault(y, x) != x:
            return False
    return True
