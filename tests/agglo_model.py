"""A literal, dictionary-based model of the agglomeration specification at the top of oracle/seg_ref.c
(waterz IterativeRegionMerging::mergeUntil / mergeRegions with OneMinus<MeanAffinity>), written independently of
the C code to cross-check it on small volumes.  Test infrastructure only; pure Python, quadratic in places."""
import numpy as np


def region_graph(affs, frags, values=False):
    """{(u, v): [sum, count]} with u < v fragment ids; channel d at the higher-index voxel of each pair
    (values=True: {(u, v): [every affinity of the edge]})."""
    edges = {}
    for d in range(3):
        lo = [slice(None)] * 3
        hi = [slice(None)] * 3
        lo[d], hi[d] = slice(0, -1), slice(1, None)
        a, b, w = frags[tuple(lo)], frags[tuple(hi)], affs[d][tuple(hi)]
        m = (a != b) & (a != 0) & (b != 0)
        for x, y, v in zip(a[m].tolist(), b[m].tolist(), w[m].tolist()):
            if values:
                edges.setdefault((min(x, y), max(x, y)), []).append(int(v))
                continue
            e = edges.setdefault((min(x, y), max(x, y)), [0, 0])
            e[0] += int(v)
            e[1] += 1
    return edges


def _quantile_score(values, q):
    """OneMinus<HistogramQuantileAffinity<., q, ., 256, .>> of a multiset of uint8 affinities (= their histogram bins):
    pivot = q * n / 100 + 1 (1-based, integer); value = (the pivot-th smallest bin + 0.5) / 256."""
    v = sorted(values)
    pivot = q * len(v) // 100 + 1
    b = v[min(pivot, len(v)) - 1] if pivot <= len(v) else 255
    return np.float32(1.0) - (np.float32(b) + np.float32(0.5)) / np.float32(256.0)


def _score(sum_, cnt):
    return np.float32(1.0) - np.float32(np.float64(sum_) / (255.0 * np.float64(cnt)))


class Merger:
    def __init__(self, affs, frags, nbins=0, quantile=None, init_with_max=False):
        """quantile None: mean scoring over (sum, count); else histogram-quantile scoring over the edges' affinity multisets
        (`sum` then holds the list of values and `cnt` is unused: merging two edges concatenates their lists)."""
        self.quantile = quantile
        g = region_graph(affs, frags, values=quantile is not None)
        self.key0 = sorted(g)                              # edge index = position in ascending (u, v) order
        self.ends = [list(k) for k in self.key0]
        if quantile is None:
            self.sum = [g[k][0] for k in self.key0]
            self.cnt = [g[k][1] for k in self.key0]
        else:
            self.sum = [[max(g[k])] if init_with_max else list(g[k]) for k in self.key0]
            self.cnt = [0 for _ in self.key0]
        ne = len(self.key0)
        self.deleted = [False] * ne
        self.stale = [False] * ne
        self.stored = [self._score(e) for e in range(ne)]
        self.parent = {}
        self.nbins = nbins
        self.history = []                                  # (a, b, score)
        if nbins:
            self.bins = [[] for _ in range(nbins)]
            for e in range(ne):
                self._push(e)
        else:
            self.queue = set(range(ne))

    def _score(self, e):
        if self.quantile is None:
            return _score(self.sum[e], self.cnt[e])
        return _quantile_score(self.sum[e], self.quantile)

    def _push(self, e):
        if self.nbins:
            b = min(max(int(self.stored[e] * np.float32(self.nbins - 1)), 0), self.nbins - 1)
            self.bins[b].append(e)
        else:
            self.queue.add(e)

    def _top(self):
        if self.nbins:
            for b in self.bins:
                if b:
                    return b[0]
            return None
        if not self.queue:
            return None
        return min(self.queue, key=lambda e: (self.stored[e], self.key0[e]))

    def _pop(self, e):
        if self.nbins:
            for b in self.bins:
                if b:
                    assert b[0] == e
                    b.pop(0)
                    return
        self.queue.remove(e)

    def _find(self, x, y):
        for e, (u, v) in enumerate(self.ends):
            if not self.deleted[e] and {u, v} == {x, y}:
                return e
        return None

    def merge_until(self, thr):
        thr = np.float32(thr)
        while True:
            e = self._top()
            if e is None or not (self.stored[e] < thr):
                break
            self._pop(e)
            if self.deleted[e]:
                continue
            if self.stale[e]:
                self.stale[e] = False
                self.stored[e] = self._score(e)
                self._push(e)
                continue
            a, b = min(self.ends[e]), max(self.ends[e])
            self.history.append((a, b, float(self.stored[e])))
            for g, (u, v) in enumerate(self.ends):
                if not self.deleted[g] and a in (u, v):
                    self.stale[g] = True
            for f, (u, v) in enumerate(list(self.ends)):
                if f == e or self.deleted[f] or b not in (u, v):
                    continue
                n = v if u == b else u
                g = self._find(a, n)
                if g is not None and self.stored[f] > self.stored[g]:
                    self.sum[g] = self.sum[g] + self.sum[f]     # numbers add, lists of values concatenate
                    self.cnt[g] += self.cnt[f]
                    self.deleted[f] = True
                    self.stale[g] = True
                    continue
                if g is not None:
                    self.sum[f] = self.sum[f] + self.sum[g]
                    self.cnt[f] += self.cnt[g]
                    self.deleted[g] = True
                self.ends[f] = [min(a, n), max(a, n)]
                self.stale[f] = True
            self.deleted[e] = True
            self.parent[b] = a

    def root(self, x):
        while x in self.parent:
            x = self.parent[x]
        return x

    def segmentation(self, frags):
        out = np.zeros_like(frags)
        for i in np.unique(frags):
            if i:
                out[frags == i] = self.root(int(i))
        return out


def agglomerate(affs, frags, thresholds, quantile=None, init_with_max=False):
    m = Merger(affs, frags, quantile=quantile, init_with_max=init_with_max)
    segs = []
    for t in thresholds:
        m.merge_until(t)
        segs.append(m.segmentation(frags))
    return segs


def merge_history(affs, frags, threshold, nbins):
    m = Merger(affs, frags, nbins)
    m.merge_until(threshold)
    return m.history
