#!/opt/conda/bin/python3.9
"""Generate watershed-fragment golden vectors by running the REFERENCE post/ws.py.

Must run under /opt/conda/bin/python3.9 (the only interpreter here that has
scikit-image: 0.18.3, scipy 1.7.1), in the build container only:
    /opt/conda/bin/python3.9 tools/gen_goldens_ws.py
The reference file is imported from where it lies (/root/reference/bootstrapper/post);
only seeded inputs and the reference's outputs are stored (tests/golden/ws_cases.npz).

Inputs are uint8 affinities [3][D][H][W]; they are converted exactly as the reference
call site does (post/watershed.py:259-262: astype(float32)/255) before
watershed_from_affinities(..., fragments_in_xy, return_seeds=True, min_seed_distance).
"""
import os
import sys
import warnings

import numpy as np

warnings.filterwarnings("ignore")
sys.path.insert(0, "/root/reference/bootstrapper/post")
import ws  # noqa: E402  (reference module)
from scipy.ndimage import gaussian_filter  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "ws_cases.npz")


def blobby(rng, shape, sigma, lo=0, hi=255):
    a = gaussian_filter(rng.random((3,) + shape), sigma=(0,) + sigma)
    a = (a - a.min()) / (a.max() - a.min())
    return (lo + a * (hi - lo)).astype(np.uint8)


def cases():
    rng = np.random.default_rng(1234)
    c = {}
    c["blobby_small"] = (blobby(rng, (4, 40, 48), (1, 3, 3)), True, 10)
    c["blobby_mid"] = (blobby(rng, (6, 96, 80), (1, 4, 4)), True, 10)
    c["blobby_128"] = (blobby(rng, (3, 128, 128), (1, 5, 5)), True, 10)
    c["blobby_msd5"] = (blobby(rng, (3, 64, 72), (1, 3, 3)), True, 5)
    c["blobby_msd3"] = (blobby(rng, (2, 50, 50), (1, 2, 2)), True, 3)
    # white noise: tie-rich distance fields, many tiny seeds
    c["noise"] = (rng.integers(0, 256, size=(3, 3, 48, 56), dtype=np.uint8), True, 10)
    # few grey levels -> large plateaus, exact ties a+b == 255 / 256 around the threshold
    q = rng.integers(0, 4, size=(3, 3, 40, 40)).astype(np.uint8)
    c["quantised"] = ((np.array([0, 127, 128, 255], dtype=np.uint8)[q]), True, 10)
    # degenerate slices: all background, all foreground (no background voxel: scipy EDT quirk)
    z = np.zeros((3, 2, 24, 30), dtype=np.uint8)
    c["all_zero"] = (z, True, 10)
    c["all_255"] = (np.full((3, 2, 24, 30), 255, dtype=np.uint8), True, 10)
    mixed = blobby(rng, (4, 32, 36), (1, 3, 3))
    mixed[:, 1] = 0
    mixed[:, 2] = 255
    c["mixed_degenerate"] = (mixed, True, 10)
    # a slice narrower than the max-filter window (reflect border wraps more than once)
    c["tiny"] = (blobby(rng, (2, 7, 9), (0, 1, 1)), True, 10)
    # high-contrast field: big objects
    c["big_objects"] = (blobby(rng, (2, 100, 100), (1, 8, 8), 60, 255), True, 10)
    # 3-D mode (fragments_in_xy=False)
    c["vol3d"] = (blobby(rng, (12, 40, 44), (2, 3, 3)), False, 10)
    c["vol3d_noise"] = (rng.integers(0, 256, size=(3, 6, 20, 22), dtype=np.uint8), False, 5)
    return c


def main():
    out = {}
    for name, (affs_u8, xy, msd) in cases().items():
        affs = affs_u8.astype(np.float32) / 255.0
        frags, max_id, seeds = ws.watershed_from_affinities(
            affs, fragments_in_xy=xy, return_seeds=True, min_seed_distance=msd)
        assert frags.max() < 2 ** 31
        out[name + "/affs"] = affs_u8
        out[name + "/frags"] = frags.astype(np.uint32)
        out[name + "/seeds"] = seeds.astype(np.uint32)
        out[name + "/meta"] = np.array([int(xy), msd, int(max_id)], dtype=np.int64)
        print(name, affs_u8.shape, "xy" if xy else "3d", "msd", msd, "max_id", int(max_id),
              "n_frags", len(np.unique(frags)) - 1)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT))


if __name__ == "__main__":
    main()
