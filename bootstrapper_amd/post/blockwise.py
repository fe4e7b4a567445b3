"""Blockwise fragment extraction and RAG edge scoring on the device.

Behavioural mirror of the two volara tasks of the reference,
/root/reference/bootstrapper/post/blockwise/watershed_frags.py:31-258 (`WatershedFrags`) and
/root/reference/bootstrapper/post/blockwise/waterz_agglom.py:41-181 (`WaterzAgglom`): same
per-block method names and semantics, with the arithmetic in libbsmi.  The reference hands blocks
to CPU worker processes through daisy and keeps the RAG in SQLite/PostgreSQL; here one process
per GPU walks its blocks and the RAG lives in host memory (`RagStore`, with an optional SQLite
export).  Volumes are addressed in voxels of the affinity array.

Documented choices where the reference depends on absent third-party code ([EXT]):
  * block id = z-major index of the block in the block grid (daisy uses a Cantor number of the
    block index): fragment ids differ from the reference's by a per-block constant only;
  * an edge is written by the block that holds the centre of mass of its smaller-id fragment
    (funlib.persistence `write_edges` keeps an edge when the position of `min(u, v)` lies in the
    block's write ROI), i.e. the block that created that fragment.
"""
import itertools

import numpy as np
import torch

from .engine import SegEngine


def shrink_blocks(total_shape, block_size):
    """Write boxes [(begin, end)] tiling `total_shape`, z-major; blocks at the upper faces are
    shrunk to the volume (`fit="shrink"`, watershed_frags.py:60, waterz_agglom.py:80)."""
    counts = [-(-int(t) // int(b)) for t, b in zip(total_shape, block_size)]
    out = []
    for idx in itertools.product(*[range(c) for c in counts]):
        begin = tuple(i * int(b) for i, b in zip(idx, block_size))
        end = tuple(min(bg + int(b), int(t)) for bg, b, t in zip(begin, block_size, total_shape))
        out.append((begin, end))
    return out


def read_with_fill(array, begin, end, lead=()):
    """array[..., begin:end] with zeros outside the array (`to_ndarray(roi, fill_value=0)`)."""
    nd = len(begin)
    shape = array.shape[-nd:]
    out = np.zeros(tuple(lead) + tuple(e - b for b, e in zip(begin, end)), dtype=array.dtype)
    src, dst = [], []
    for b, e, n in zip(begin, end, shape):
        lo, hi = max(b, 0), min(e, n)
        if hi <= lo:
            return out
        src.append(slice(lo, hi))
        dst.append(slice(lo - b, hi - b))
    pre = (slice(None),) * len(lead)
    out[pre + tuple(dst)] = array[pre + tuple(src)]
    return out


class RagStore:
    """Fragment nodes {position, size} and edges {merge_score} of the whole volume."""

    def __init__(self):
        self.node_ids, self.node_pos, self.node_size = [], [], []
        self.edges, self.scores = [], []

    def add_nodes(self, ids, positions, sizes):
        self.node_ids.append(np.asarray(ids, dtype=np.uint64))
        self.node_pos.append(np.asarray(positions, dtype=np.float64).reshape(-1, 3))
        self.node_size.append(np.asarray(sizes, dtype=np.int64))

    def add_edges(self, edges, scores):
        self.edges.append(np.asarray(edges, dtype=np.uint64).reshape(-1, 2))
        self.scores.append(np.asarray(scores, dtype=np.float32))

    def nodes(self):
        ids = np.concatenate(self.node_ids) if self.node_ids else np.zeros(0, np.uint64)
        order = np.argsort(ids, kind="stable")
        pos = np.concatenate(self.node_pos)[order] if self.node_pos else np.zeros((0, 3))
        size = np.concatenate(self.node_size)[order] if self.node_size else np.zeros(0, np.int64)
        return ids[order], pos, size

    def scored_edges(self):
        """Edges whose fragments merged in their block (`merge_score is not None`, post/watershed.py:163-171)."""
        if not self.edges:
            return np.zeros((0, 2), np.uint64), np.zeros(0, np.float32)
        e, s = np.concatenate(self.edges), np.concatenate(self.scores)
        keep = ~np.isnan(s)
        return e[keep], s[keep]

    def all_edges(self):
        if not self.edges:
            return np.zeros((0, 2), np.uint64), np.zeros(0, np.float32)
        return np.concatenate(self.edges), np.concatenate(self.scores)

    def to_sqlite(self, path):
        """Export (nodes: id, z, y, x, size; edges: u, v, merge_score with NULL = never merged)."""
        import os
        import sqlite3
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        # natively (csrc/rag_db.cpp: prepared statements through the SQLite C API, rows in key order) when libsqlite3 can be
        # loaded; else through the interpreter's module, a tuple per row (3x slower on a million rows)
        import ctypes as C
        from .. import _lib
        ids, pos, size = self.nodes()
        e, s = self.all_edges()
        ids = np.ascontiguousarray(ids, np.uint64)
        pos = np.ascontiguousarray(pos, np.float64).reshape(-1, 3)
        size = np.ascontiguousarray(size, np.int64)
        e = np.ascontiguousarray(e, np.uint64).reshape(-1, 2)
        s = np.ascontiguousarray(s, np.float32)
        rc = _lib.lib.bsmi_rag_write_sqlite(os.fsencode(path), len(ids), ids.ctypes.data, pos.ctypes.data, size.ctypes.data, len(e), e.ctypes.data, s.ctypes.data)
        if rc == 0:
            return
        if rc != _lib.ERR_MISSING:
            _lib.check(rc)
        con = sqlite3.connect(path)
        # a file written once from scratch: no rollback journal, no fsync per transaction (a million rows: 6 s -> under 2)
        con.execute("PRAGMA journal_mode = OFF")
        con.execute("PRAGMA synchronous = OFF")
        con.execute("PRAGMA cache_size = -400000")
        con.execute("PRAGMA locking_mode = EXCLUSIVE")
        with con:
            con.execute("DROP TABLE IF EXISTS nodes")
            con.execute("DROP TABLE IF EXISTS edges")
            con.execute("CREATE TABLE nodes (id INTEGER PRIMARY KEY, z REAL, y REAL, x REAL, size INTEGER)")
            con.execute("CREATE TABLE edges (u INTEGER, v INTEGER, merge_score REAL, PRIMARY KEY (u, v)) WITHOUT ROWID")
            ids, pos, size = self.nodes()
            # rows as tuples of Python numbers through numpy's own conversion (`tolist`), ids in ascending order (sequential
            # inserts into the primary-key tree); 64-bit ids above 2^63 - 1 do not occur (block id * 2^21 + label)
            order = np.argsort(ids, kind="stable")
            pos = np.asarray(pos, np.float64)[order]
            con.executemany("INSERT INTO nodes VALUES (?, ?, ?, ?, ?)",
                            zip(ids[order].astype(np.int64).tolist(), pos[:, 0].tolist(), pos[:, 1].tolist(), pos[:, 2].tolist(),
                                np.asarray(size, np.int64)[order].tolist()))
            e, s = self.all_edges()
            e = np.asarray(e, np.uint64).reshape(-1, 2).astype(np.int64)
            order = np.lexsort((e[:, 1], e[:, 0]))
            sc = np.asarray(s, np.float64)[order].astype(object)
            sc[np.isnan(np.asarray(s, np.float64)[order])] = None
            con.executemany("INSERT OR REPLACE INTO edges VALUES (?, ?, ?)", zip(e[order, 0].tolist(), e[order, 1].tolist(), sc.tolist()))
        con.close()


class _BlockTask:
    def __init__(self, block_size, context, total_shape, device=0, origin=(0, 0, 0)):
        self.origin = tuple(int(o) for o in origin)  # voxel offset of the ROI inside the affinity array
        self.block_size = tuple(int(b) for b in block_size)
        self.context = tuple(int(c) for c in context)
        self.total_shape = tuple(int(t) for t in total_shape)
        self.device = int(device)
        self.blocks = shrink_blocks(self.total_shape, self.block_size)
        read_shape = tuple(min(b, t) + 2 * c for b, t, c in zip(self.block_size, self.total_shape, self.context))
        self.engine = SegEngine(read_shape, self.device)

    @property
    def num_voxels_in_block(self):
        return int(np.prod(self.block_size))

    def read_box(self, write_box):
        begin, end = write_box
        return tuple(b - c for b, c in zip(begin, self.context)), tuple(e + c for e, c in zip(end, self.context))

    def read_array(self, array, rbegin, rend, lead=()):
        """ROI-relative read box -> data of the (larger) array, zeros outside it."""
        return read_with_fill(array, tuple(b + o for b, o in zip(rbegin, self.origin)),
                              tuple(e + o for e, o in zip(rend, self.origin)), lead)


class WatershedFrags(_BlockTask):
    """Per-block fragments with context (reference watershed_frags.py:31-258)."""

    def __init__(self, block_size, context, total_shape, fragments_in_xy=True, min_seed_distance=10, seed_eps=None,
                 epsilon_agglomerate=0.0, sigma=None, noise_eps=None, bias=None, filter_fragments=0.0,
                 remove_debris=0, device=0, origin=(0, 0, 0)):
        super().__init__(block_size, context, total_shape, device, origin)
        self.epsilon_agglomerate = float(epsilon_agglomerate or 0.0)
        self.shift = dict(sigma=sigma, noise_eps=noise_eps, bias=bias, seed_eps=seed_eps)
        self.fragments_in_xy = bool(fragments_in_xy)
        self.min_seed_distance = int(min_seed_distance)
        self.filter_fragments = float(filter_fragments)
        self.remove_debris = int(remove_debris)

    def watershed_in_block(self, block_index, affs, frags_out, rag, offset=(0, 0, 0), voxel_size=(1, 1, 1), mask=None):
        """watershed_frags.py:196-246.  affs: u8 array-like [C][Z][Y][X] (numpy / ZarrArray view of
        the ROI); frags_out: uint64 numpy volume of the ROI, the write box is filled in."""
        wbox = self.blocks[block_index]
        rbegin, rend = self.read_box(wbox)
        a = self.read_array(affs, rbegin, rend, lead=(affs.shape[0],))[:3]
        if a.shape[0] == 2:
            a = np.concatenate([np.zeros_like(a[:1]), a])
        if a.max() == 0:  # `affs_data.max() < 1e-3` on uint8 data
            return 0
        dev = torch.device("cuda", self.device)
        a_dev = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        if mask is not None:
            m = self.read_array(mask, rbegin, rend)
            a_dev = a_dev * torch.from_numpy((m > 0).astype(np.uint8)).to(dev)
        eng = self.engine
        src = a_dev
        if any(v is not None for v in self.shift.values()):  # compute_fragments, watershed_frags.py:116-145
            from .shifts import boundary_mask_affinities
            src = boundary_mask_affinities(a_dev, self.fragments_in_xy, dtype=torch.float64,
                                           generator=torch.Generator(device=dev).manual_seed(block_index),
                                           min_seed_distance=self.min_seed_distance, **self.shift)
        frags, _ = eng.ws_fragments(src, self.fragments_in_xy, self.min_seed_distance)
        if self.epsilon_agglomerate > 0:                     # epsilon_agglomerate_fragments, :158-177
            eng.rag_agglomerate(a_dev.contiguous(), frags, self.epsilon_agglomerate, 256)
        crop_off = tuple(w - r for w, r in zip(wbox[0], rbegin))
        crop_shape = tuple(e - b for b, e in zip(*wbox))
        id_offset = block_index * self.num_voxels_in_block
        labels, num = eng.postprocess_fragments(a_dev, frags, self.filter_fragments, self.remove_debris, crop_off,
                                                crop_shape, id_offset)
        eng.status()
        n = int(num.item())
        assert n < self.num_voxels_in_block, f"max_id: {n}"
        sl = tuple(slice(b, e) for b, e in zip(*wbox))
        frags_out[sl] = labels.cpu().numpy().view(np.uint64)
        if n == 0:
            return 0
        size, sums = eng.label_stats(labels, id_offset, n)
        size_h = size.cpu().numpy()
        centre = sums.cpu().numpy().astype(np.float64) / size_h[:, None]
        position = (np.asarray(offset, np.float64) + (np.asarray(wbox[0], np.float64) + centre) * np.asarray(voxel_size, np.float64))
        rag.add_nodes(np.arange(1, n + 1, dtype=np.uint64) + np.uint64(id_offset), position, size_h)
        return n


class WaterzAgglom(_BlockTask):
    """Per-block RAG edge scoring (reference waterz_agglom.py:41-181)."""

    def __init__(self, block_size, context, total_shape, merge_function="mean", discretize_queue=256, device=0,
                 origin=(0, 0, 0)):
        super().__init__(block_size, context, total_shape, device, origin)
        if merge_function != "mean":
            raise NotImplementedError(f"merge_function {merge_function!r}: only 'mean' is implemented "
                                      "(the one the reference enables, waterz_agglom.py:25)")
        self.discretize_queue = int(discretize_queue)

    def agglomerate_in_block(self, block_index, affs, frags, rag):
        """waterz_agglom.py:106-170; keeps the edges this block owns (see the module docstring)."""
        wbox = self.blocks[block_index]
        rbegin, rend = self.read_box(wbox)
        a = self.read_array(affs, rbegin, rend, lead=(affs.shape[0],))[:3]
        if a.shape[0] == 2:
            a = np.concatenate([np.zeros_like(a[:1]), a])
        f = read_with_fill(frags, rbegin, rend)
        if not f.any():
            return 0
        dev = torch.device("cuda", self.device)
        edges, scores = self.engine.rag_merge_scores(torch.from_numpy(np.ascontiguousarray(a)).to(dev),
                                                     torch.from_numpy(np.ascontiguousarray(f).view(np.int64)).to(dev),
                                                     1.0, self.discretize_queue)
        e = edges.cpu().numpy().view(np.uint64)
        s = scores.cpu().numpy()
        nv = np.uint64(self.num_voxels_in_block)
        own = (e[:, 0] - np.uint64(1)) // nv == np.uint64(block_index)
        rag.add_edges(e[own], s[own])
        return int(own.sum())
