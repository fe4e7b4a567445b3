import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bootstrapper_amd.zarr_io import open_ds, prepare_ds
from bootstrapper_amd.volume import SlabSegmenter
import bootstrapper_amd.post.watershed as W
rng = np.random.default_rng(1)
affs = rng.integers(1, 255, size=(3, 20, 150, 130), dtype=np.uint8)
store = tempfile.mkdtemp() + "/v.zarr"
ds = prepare_ds(store + "/affs", affs.shape, chunk_shape=(3, 8, 64, 64), dtype=np.uint8, axis_names=["c^", "z", "y", "x"], compressor="zlib", voxel_size=(1, 1, 1), offset=(0, 0, 0))
ds[:] = affs
a = open_ds(store + "/affs")
for (z0, nzs) in ((0, 16), (8, 12), (16, 4), (16, 4), (12, 8), (15, 5)):
    seg = SlabSegmenter((nzs, 150, 130), (8, 64, 64), (1, 8, 8), 3, z0 // 8, [0.3], True, 4, 0.35, 12, 256, n_lanes=4, device=0, exchange_affs=False, total_rows=3, lazy_outputs=True)
    W._fill_affinities(seg, a, (0, 0, 0), z0, None, 0)
    torch.cuda.synchronize()
    got = seg.interior(seg.affs).cpu().numpy()
    print(z0, nzs, "equal" if np.array_equal(got, affs[:, z0:z0 + nzs]) else "DIFFERENT", int((got > 0).sum()), flush=True)
    del seg
