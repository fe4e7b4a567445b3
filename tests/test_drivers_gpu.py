"""End to end through the drivers on a small Zarr volume: `run_prediction` (blockwise, reflect
padded, uint8 store) then `run_segmentation` (fragments + agglomeration datasets), each compared
with the oracle applied to the same data.  Needs an MI355X."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(tmp_path, golden_dir):
    from bootstrapper_amd.zarr_io import prepare_ds
    d = np.load(os.path.join(golden_dir, "unet_affs_f4i2.npz"))
    sd = {k[2:]: d[k] for k in d.files if k.startswith("w:")}
    setup = tmp_path / "setup_01"
    setup.mkdir()
    nc = {"in_channels": 1, "num_fmaps": 4, "fmap_inc_factor": 2, "downsample_factors": [[1, 2, 2]] * 3,
          "kernel_size_down": [[[3, 3, 3], [3, 3, 3]]] * 4, "kernel_size_up": [[[3, 3, 3], [3, 3, 3]]] * 3,
          "input_shape": [30, 108, 108], "output_shape": [2, 16, 16], "shape_increase": [8, 16, 16],
          "inputs": {"raw": {"dims": 1}}, "outputs": {"3d_affs": {"dtype": "uint8", "dims": 6}}}
    (setup / "net_config.json").write_text(json.dumps(nc))
    ckpt = str(setup / "model_checkpoint_1000")
    torch.save({"state_dict": {"model." + k: torch.from_numpy(v) for k, v in sd.items()}}, ckpt + ".ckpt")
    rng = np.random.default_rng(3)
    raw = rng.integers(0, 256, size=(23, 50, 61), dtype=np.uint8)
    store = str(tmp_path / "vol.zarr")
    ds = prepare_ds(store + "/raw", raw.shape, offset=(80, 8, 8), voxel_size=(40, 4, 4), chunk_shape=(8, 32, 32),
                    dtype=np.uint8, axis_names=["z", "y", "x"], units=["nm"] * 3)
    ds[:] = raw
    cfg = tmp_path / "pred.toml"
    cfg.write_text(f'''["01-3d_affs"]
setup_dir = "{setup}"
input_datasets = ["{store}/raw"]
checkpoint = "{ckpt}"
output_datasets_prefix = "{store}/predictions"
chain_str = ""
num_workers = 1
num_gpus = 1
''')
    return nc, sd, raw, store, str(cfg)


def test_predict_then_segment_drivers(tmp_path, golden_dir):
    from bootstrapper_amd.predict import run_prediction
    from bootstrapper_amd.segment import run_segmentation
    from bootstrapper_amd.zarr_io import open_ds
    from oracle import unet_ref as R
    from oracle import seg_ref as S
    nc, sd, raw, store, cfg = _setup(tmp_path, golden_dir)
    run_prediction(cfg, "01", precision="f32")
    out = open_ds(store + "/predictions/1000/3d_affs")
    assert out.shape == (6, 23, 50, 61) and out.chunks == (6, 10, 32, 32) and out.dtype == np.uint8
    assert out.offset == (80, 8, 8) and out.voxel_size == (40, 4, 4) and out.axis_names == ["c^", "z", "y", "x"]
    got = out[:]
    # oracle: reflect pad the dataset, predict every block, clip to the ROI
    ctx = (14, 46, 46)
    full = np.pad(raw, [(c, c + 64) for c in ctx], mode="reflect")
    ref = np.zeros_like(got)
    ocfg = R.default_cfg(4, 2)
    ob = (10, 32, 32)
    for z in range(0, 23, ob[0]):
        for y in range(0, 50, ob[1]):
            for x in range(0, 61, ob[2]):
                blk = full[z:z + 38, y:y + 124, x:x + 124]
                o = R.to_u8(R.predict_block(ocfg, sd, blk, ["affs_head"])[0])
                hz, hy, hx = min(ob[0], 23 - z), min(ob[1], 50 - y), min(ob[2], 61 - x)
                ref[:, z:z + hz, y:y + hy, x:x + hx] = o[:, :hz, :hy, :hx]
    diff = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    assert diff.max() <= 1 and (diff == 0).mean() > 0.99

    seg_cfg = tmp_path / "seg.toml"
    seg_cfg.write_text(f'''affs_dataset = "{store}/predictions/1000/3d_affs"
fragments_dataset = "{store}/fragments"
seg_dataset_prefix = "{store}/segmentations"
blockwise = false
[ws_params]
thresholds = [0.3, 0.6]
min_seed_distance = 4
''')
    written = run_segmentation(str(seg_cfg), "ws")
    assert [os.path.relpath(w, store) for w in written] == [
        "fragments/xy--msd4", "segmentations/mfmean--t0.3--xy--msd4", "segmentations/mfmean--t0.6--xy--msd4"]
    frags_ref, _ = S.ws_fragments_u8(got[:3], True, 4)
    segs_ref = S.agglomerate_mean_u8(got[:3], frags_ref, [0.3, 0.6])
    f = open_ds(written[0])
    assert f.dtype == np.uint64 and f.offset == (80, 8, 8) and f.axis_names == ["z", "y", "x"]
    assert f.attrs["bs_params"]["method"] == "ws" and f.attrs["bs_params"]["blockwise"] is False
    assert np.array_equal(f[:], frags_ref)
    for w, r in zip(written[1:], segs_ref):
        assert np.array_equal(open_ds(w)[:], r)
    with pytest.raises(ValueError, match="Blockwise requires a database config"):
        run_segmentation(str(seg_cfg), "ws", blockwise=True, param=())
    from bootstrapper_amd.post.watershed import watershed_segmentation
    with pytest.raises(NotImplementedError):
        watershed_segmentation({"blockwise": True})
