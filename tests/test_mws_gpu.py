"""`bs segment --mws`: the device mutex watershed against the numpy restatement (oracle/mws_ref.py; parity unpinned:
mwatershed and volara are third party and absent, see the oracle's header)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NBHD = [[-1, 0, 0], [0, -1, 0], [0, 0, -1], [-2, 0, 0], [0, -3, 0], [0, 0, -3]]
BIAS = [-0.4, -0.4, -0.4, -0.7, -0.7, -0.7]


def _smooth_affs(shape, seed, k=len(NBHD)):
    """Affinity-like data with structure (a few blobs) plus noise, in [0, 1]."""
    rng = np.random.default_rng(seed)
    zz, yy, xx = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
    lab = np.zeros(shape, np.int64)
    for i in range(6):
        c = [rng.integers(0, s) for s in shape]
        r = rng.integers(2, max(3, min(shape) // 2))
        lab[(zz - c[0]) ** 2 + (yy - c[1]) ** 2 + (xx - c[2]) ** 2 < r * r] = i + 1
    a = np.zeros((k,) + tuple(shape))
    for c, (oz, oy, ox) in enumerate(NBHD[:k]):
        shifted = np.roll(lab, (-oz, -oy, -ox), axis=(0, 1, 2))
        a[c] = (lab == shifted) & (lab > 0)
    a = 0.8 * a + 0.1 + 0.08 * rng.standard_normal(a.shape)
    return np.clip(a, 0, 1)


@pytest.mark.parametrize("shape,seed,strides", [
    ((6, 9, 10), 0, None),
    ((5, 12, 12), 1, [[1, 1, 1]] * 3 + [[1, 2, 2]] * 3),
    ((1, 16, 16), 2, None),
    ((8, 8, 8), 3, [[1, 1, 1]] * 3 + [[2, 3, 3]] * 3),
])
def test_grid_mutex_watershed_matches_restatement(shape, seed, strides):
    import torch
    from oracle import mws_ref
    from bootstrapper_amd.post.mws import mws_agglom
    a = _smooth_affs(shape, seed) + np.array(BIAS).reshape(-1, 1, 1, 1)
    got = mws_agglom(torch.from_numpy(a).cuda(), NBHD, strides).cpu().numpy().view(np.uint64)
    ref = mws_ref.mws_agglom(a, NBHD, strides)
    assert np.array_equal(got, ref)   # same labelling rule (1 + smallest voxel index), not just the same partition
    assert len(np.unique(got)) > 1


def test_ties_zero_and_nan_edges():
    """Quantised affinities give many equal weights (tie order = (channel, voxel) ascending); zeros and NaNs are no edges."""
    import torch
    from oracle import mws_ref
    from bootstrapper_amd.post.mws import mws_agglom
    rng = np.random.default_rng(7)
    a = rng.integers(0, 5, size=(6, 4, 10, 10)).astype(np.float64) / 4.0 - 0.5
    a[0, 1, 2, 3] = np.nan
    got = mws_agglom(torch.from_numpy(a).cuda(), NBHD).cpu().numpy().view(np.uint64)
    ref = mws_ref.mws_agglom(np.nan_to_num(a, nan=0.0), NBHD)
    assert np.array_equal(got, ref)


def test_graph_mutex_watershed_and_pair_affinities():
    import torch
    from oracle import mws_ref
    from bootstrapper_amd.post.mws import mws_cluster
    from bootstrapper_amd.post.watershed_mutex import pair_affinities
    rng = np.random.default_rng(11)
    n = 60
    e = rng.integers(0, n, size=(400, 2))
    e = e[e[:, 0] != e[:, 1]]
    s = np.round(rng.standard_normal(len(e)), 1)   # ties and exact zeros included
    assert np.array_equal(mws_cluster(n, e, s), mws_ref.mws_cluster(n, e, s))
    assert np.array_equal(mws_cluster(5, np.zeros((0, 2)), np.zeros(0)), np.arange(1, 6, dtype=np.uint64))
    with pytest.raises(RuntimeError):
        mws_cluster(3, [[0, 7]], [1.0])

    shape = (6, 10, 11)
    frags = rng.integers(0, 9, size=shape).astype(np.uint64) * np.uint64(1 << 33)   # ids beyond 32 bits, background 0
    affs = rng.integers(0, 256, size=(len(NBHD),) + shape).astype(np.uint8)
    e, mean, cnt = pair_affinities(torch.from_numpy(affs).cuda(), NBHD, torch.from_numpy(frags.view(np.int64)).cuda())
    ref = mws_ref.pair_affinity(affs, NBHD, frags)
    assert len(e) == len(ref) > 0
    for (u, v), m, c in zip(e.tolist(), mean.tolist(), cnt.tolist()):
        rs, rc = ref[(u, v)]
        assert c == rc and m == rs / rc / 255.0


def test_mwatershed_from_affinities_mirror():
    """post/mws.py surface: numpy in / numpy out, bias + sigma shift, reproducible noise with a seed."""
    from oracle import mws_ref
    from bootstrapper_amd.post.mws import mwatershed_from_affinities
    from scipy.ndimage import gaussian_filter
    a = _smooth_affs((4, 12, 12), 5)
    sigma = [0, 1.0, 1.0]
    got = mwatershed_from_affinities(a, NBHD, BIAS, sigma=sigma)
    shift = gaussian_filter(a, sigma=(0, *sigma)) - a + np.array(BIAS).reshape(-1, 1, 1, 1)
    ref = mws_ref.mws_agglom(a + shift, NBHD)
    assert got.dtype == np.uint64 and mws_ref.same_partition(got, ref)
    n1 = mwatershed_from_affinities(a, NBHD, BIAS, noise_eps=0.01, seed=3)
    n2 = mwatershed_from_affinities(a, NBHD, BIAS, noise_eps=0.01, seed=3)
    assert np.array_equal(n1, n2)
    with pytest.raises(ValueError):
        mwatershed_from_affinities(a, NBHD[:3], BIAS)


def _write_affs(tmp_path, a_u8):
    from bootstrapper_amd.zarr_io import prepare_ds
    path = os.path.join(str(tmp_path), "vol.zarr", "affs")
    ds = prepare_ds(path, shape=a_u8.shape, offset=(0, 0, 0), voxel_size=(1, 1, 1), axis_names=["c^", "z", "y", "x"], units=["nm"] * 3,
                    dtype=np.uint8, chunk_shape=(a_u8.shape[0], 8, 16, 16))
    ds[:] = a_u8
    return path


def test_simple_mutex_and_blockwise_drivers(tmp_path):
    """`bs segment --mws` through run_segmentation: simple (whole ROI) against the restatement; blockwise with one block
    equals the stage-by-stage restatement (fragments, clean-up, pair affinities, graph clustering, relabel); several blocks
    with context give a consistent volume (every segment is a union of whole fragments)."""
    from oracle import mws_ref
    from scipy import ndimage
    from bootstrapper_amd.zarr_io import open_ds
    from bootstrapper_amd.post.watershed_mutex import mutex_watershed_segmentation
    a = (_smooth_affs((8, 32, 32), 9) * 255).astype(np.uint8)
    path = _write_affs(tmp_path, a)
    base = dict(affs_dataset=path, fragments_dataset=os.path.join(str(tmp_path), "vol.zarr", "fragments"),
                seg_dataset_prefix=os.path.join(str(tmp_path), "vol.zarr", "segmentations"), lut_dir=os.path.join(str(tmp_path), "luts"),
                aff_neighborhood=NBHD, bias=BIAS, sigma=None, noise_eps=None, strides=None, randomized_strides=False, remove_debris=4,
                filter_fragments=0.1, min_seed_distance=None, global_bias=[1.0, -0.5])
    frags_name, seg_name = mutex_watershed_segmentation(dict(base))
    ref = mws_ref.mws_agglom(a.astype(np.float64) / 255.0 + np.array(BIAS).reshape(-1, 1, 1, 1), NBHD)
    got = open_ds(frags_name)[:]
    assert mws_ref.same_partition(got, ref)
    seg = open_ds(seg_name)[:]
    ids, counts = np.unique(ref, return_counts=True)
    small = np.isin(ref, ids[counts < 4])
    assert np.array_equal(seg == 0, small) and mws_ref.same_partition(seg[~small], ref[~small])
    assert open_ds(seg_name).attrs["bs_params"]["method"] == "mws"

    # blockwise, one block covering the ROI (block_shape "roi"): stage by stage against the restatement
    cfg = dict(base, blockwise=True, block_shape="roi", db={"db_file": os.path.join(str(tmp_path), "rag.db")})
    frags_name, seg_name = mutex_watershed_segmentation(cfg)
    fr = open_ds(frags_name)[:]
    # clean-up of the reference fragments: mean affinity of the first three channels >= 0.1, debris of < 4 voxels, 26-connected relabel
    mean3 = a[:3].astype(np.float64).mean(axis=0) / 255.0
    keep = ref.copy()
    for i in ids:
        m = ref == i
        if mean3[m].mean() < 0.1:
            keep[m] = 0
    for i, c in zip(ids, counts):                      # debris: whole fragments of fewer than 4 voxels
        if c < 4:
            keep[ref == i] = 0
    # 26-connected components of EQUAL labels (skimage.measure.label on the label volume)
    parts = np.zeros(ref.shape, np.int64)
    nxt = 0
    for i in np.unique(keep[keep > 0]):
        l, k = ndimage.label(keep == i, structure=np.ones((3, 3, 3)))
        parts[l > 0] = l[l > 0] + nxt
        nxt += k
    assert mws_ref.same_partition(fr, parts)
    pa = mws_ref.pair_affinity(a, NBHD, fr)
    fid = np.unique(fr[fr > 0])
    edges = np.array(sorted(pa), dtype=np.uint64).reshape(-1, 2)
    scores = np.array([pa[tuple(e)][0] / pa[tuple(e)][1] / 255.0 for e in edges.tolist()], dtype=np.float32).astype(np.float64) - 0.5
    lab_ref = mws_ref.mws_cluster(len(fid), np.searchsorted(fid, edges), scores)
    seg_ref = np.zeros_like(fr)
    seg_ref[fr > 0] = fid[lab_ref.astype(np.int64) - 1][np.searchsorted(fid, fr[fr > 0])]
    assert np.array_equal(open_ds(seg_name)[:], seg_ref)
    assert os.path.exists(os.path.join(str(tmp_path), "rag.db"))

    # several blocks with context
    cfg = dict(base, blockwise=True, block_shape=[8, 16, 16], context=[2, 4, 4], db={"db_file": os.path.join(str(tmp_path), "rag2.db")},
               fragments_dataset=os.path.join(str(tmp_path), "vol.zarr", "fragments_b"),
               seg_dataset_prefix=os.path.join(str(tmp_path), "vol.zarr", "segmentations_b"))
    frags_name, seg_name = mutex_watershed_segmentation(cfg)
    fr, sg = open_ds(frags_name)[:], open_ds(seg_name)[:]
    assert fr.max() > 8 * 16 * 16 and np.array_equal(fr == 0, sg == 0)
    f_ids, first = np.unique(fr[fr > 0], return_index=True)
    lut = dict(zip(f_ids.tolist(), sg[fr > 0][first].tolist()))
    assert np.array_equal(sg[fr > 0], np.array([lut[i] for i in fr[fr > 0].tolist()], dtype=np.uint64))
    assert len(np.unique(sg[sg > 0])) < len(f_ids)   # the graph clustering merged fragments across blocks
