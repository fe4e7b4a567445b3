#!/opt/conda/bin/python3.9
"""Golden vectors for the third-party primitives the reference's blockwise fragment task calls
(/root/reference/bootstrapper/post/blockwise/watershed_frags.py:148-156,188-192,222):
  scipy.ndimage.mean(average_affs, fragments, ids)          -> filter_avg_fragments
  skimage.morphology.remove_small_objects(frags, min_size)  -> remove_debris
  skimage.measure.label(frags, return_num=True)             -> relabel after the crop
watershed_frags.py itself cannot be imported here (volara / funlib are absent), so these are
the library calls exactly as that file makes them, on seeded inputs, under the only interpreter
that has scikit-image (0.18.3).  Run in the build container only:
    /opt/conda/bin/python3.9 tools/gen_goldens_blockwise.py   (writes tests/golden/blockwise_cases.npz)
"""
import os
import warnings

import numpy as np

warnings.filterwarnings("ignore")
from scipy.ndimage import mean as ndi_mean, gaussian_filter  # noqa: E402
from skimage.measure import label as relabel  # noqa: E402
from skimage.morphology import remove_small_objects  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "blockwise_cases.npz")
rng = np.random.default_rng(77)
out = {}


def blobby_labels(shape, n):
    """random label volume with blobby regions, 0 = background, ids sparse and unordered"""
    f = gaussian_filter(rng.random((n,) + shape), sigma=(0, 1, 2, 2))
    lab = np.argmax(f, axis=0).astype(np.uint64)
    ids = rng.choice(np.arange(1, 5000), size=n, replace=False).astype(np.uint64)
    ids[0] = 0
    return ids[lab]


for name, shape, n in [("a", (6, 20, 24), 12), ("b", (10, 32, 28), 40), ("c", (4, 16, 16), 5)]:
    frags = blobby_labels(shape, n)
    affs = (gaussian_filter(rng.random((3,) + shape), sigma=(0, 1, 2, 2)) * 2 - 0.5).clip(0, 1)
    affs_u8 = (affs * 255).astype(np.uint8)
    # filter_avg_fragments (float64 path of watershed_in_block: u8 -> f64 / 255)
    a64 = affs_u8.astype(np.float64) / 255.0
    average = np.mean(a64[0:3], axis=0)
    ids = np.unique(frags)
    means = ndi_mean(average, frags, ids)
    out[name + "/frags"] = frags
    out[name + "/affs"] = affs_u8
    out[name + "/ids"] = ids
    out[name + "/means"] = np.asarray(means, dtype=np.float64)
    for thr in (0.1, 0.35, 0.5):
        filtered = np.array([f for f, m in zip(ids, means) if m < thr], dtype=np.uint64)
        out[name + f"/filtered_{thr}"] = filtered
    for ms in (1, 8, 40, 200):
        kept = remove_small_objects(frags.astype(np.int64), min_size=ms).astype(np.uint64)
        out[name + f"/debris_{ms}"] = kept
    # relabel of a crop (the block's write ROI), full connectivity, raster order
    crop = frags[1:-1, 2:-3, 3:-2]
    lab, num = relabel(crop, return_num=True)
    out[name + "/crop_label"] = lab.astype(np.uint32)
    out[name + "/crop_num"] = np.array([num], dtype=np.int64)

# labels that touch only diagonally / a checkerboard: exercises 26-connectivity and ordering
cb = np.indices((4, 6, 6)).sum(axis=0) % 2
cb = (cb * 7).astype(np.uint64)
lab, num = relabel(cb, return_num=True)
out["checker/frags"] = cb
out["checker/label"] = lab.astype(np.uint32)
out["checker/num"] = np.array([num], dtype=np.int64)
two = np.zeros((3, 5, 5), np.uint64)
two[0, 0, 0] = 9; two[1, 1, 1] = 9; two[2, 3, 3] = 9; two[0, 4, 4] = 3; two[0, 3, 4] = 3
lab, num = relabel(two, return_num=True)
out["diag/frags"] = two
out["diag/label"] = lab.astype(np.uint32)
out["diag/num"] = np.array([num], dtype=np.int64)

np.savez_compressed(OUT, **out)
print("wrote", OUT, os.path.getsize(OUT), "keys", len(out))
