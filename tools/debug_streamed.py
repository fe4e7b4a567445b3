"""Dev tool: the streamed (out-of-HBM) segmentation against the resident one on the driver test's volume; prints where they differ."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.ndimage import gaussian_filter
from bootstrapper_amd.segment import run_segmentation
from bootstrapper_amd.zarr_io import open_ds, prepare_ds
from bootstrapper_amd.volume import SlabSegmenter
tmp = tempfile.mkdtemp()
rng = np.random.default_rng(21)
shape = (20, 150, 130)
a = gaussian_filter(rng.random((3,) + shape), sigma=(0, 1, 3, 3))
affs = ((a - a.min()) / (a.max() - a.min()) * 255).astype(np.uint8)
affs[:, :, :40, :50] = 0
store = tmp + "/vol.zarr"
ds = prepare_ds(store + "/affs", affs.shape, offset=(40, 8, 16), voxel_size=(40, 4, 4), chunk_shape=(3, 8, 64, 64), dtype=np.uint8,
                axis_names=["c^", "z", "y", "x"], units=["nm"] * 3, compressor="zlib")
ds[:] = affs
base = f'''affs_dataset = "{store}/affs"
fragments_dataset = "{store}/fragmentsNAME"
seg_dataset_prefix = "{store}/segmentationsNAME"
blockwise = true
EXTRA
[db]
db_file = "TMP/ragNAME.db"
[ws_params]
thresholds = [0.3, 0.45]
min_seed_distance = 4
filter_fragments = 0.35
remove_debris = 12
'''
need = [SlabSegmenter.hbm_bytes((min(nl * 8, 20), 150, 130), (1, 8, 8), 2, nl * 9) for nl in (2, 3)]
outs = {}
for name, extra in (("res", ""), ("st", f"hbm_budget_gb = {(need[0] + need[1]) / 2 / 2**30:.6f}")):
    cfg = tmp + f"/{name}.toml"
    open(cfg, "w").write(base.replace("NAME", "_" + name).replace("EXTRA", extra).replace("TMP", tmp))
    outs[name] = run_segmentation(cfg, "ws")
for a_, b_ in zip(outs["res"], outs["st"]):
    x, y = open_ds(a_)[:], open_ds(b_)[:]
    d = x != y
    print(os.path.basename(os.path.dirname(a_)), "differs" if d.any() else "equal", int(d.sum()))
    if d.any():
        for z in range(x.shape[0]):
            if d[z].any():
                idx = np.argwhere(d[z])
                print("  z", z, "n", int(d[z].sum()), "y", idx[:, 0].min(), idx[:, 0].max(), "x", idx[:, 1].min(), idx[:, 1].max(), "e.g.", x[z][tuple(idx[0])], y[z][tuple(idx[0])])

# block_shape = "roi" with and without the warm-up thread, against the CPU composition
from oracle.blockwise_ref import cpu_blockwise
fr2, _, _, _, segs2 = cpu_blockwise(affs, shape, (0, 0, 0), 4, 0.35, 12, [0.3, 0.45])
for warm in ("1", "0"):
    os.environ["BSMI_SEG_WARM"] = warm
    cfg = tmp + f"/roi{warm}.toml"
    open(cfg, "w").write(base.replace("NAME", "_roi" + warm).replace("EXTRA", 'block_shape = "roi"').replace("TMP", tmp))
    out = run_segmentation(cfg, "ws")
    x = open_ds(out[0])[:]
    d = x != fr2
    print("roi warm", warm, "fragments", "differ" if d.any() else "equal", int(d.sum()), "nonzero got/ref", int((x > 0).sum()), int((fr2 > 0).sum()), "max", x.max(), fr2.max())
    if d.any():
        zs = [z for z in range(x.shape[0]) if d[z].any()]
        print("   z with differences", zs[:30])
