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
    # one of the histogram-quantile merge functions of the non-blockwise path (post/watershed.py:230-243), `-p` style override
    written_h = run_segmentation(str(seg_cfg), "ws", param=("merge_function=hist_quant_75",))
    assert [os.path.relpath(w, store) for w in written_h][1:] == ["segmentations/mfhist_quant_75--t0.3--xy--msd4",
                                                                  "segmentations/mfhist_quant_75--t0.6--xy--msd4"]
    for w, r in zip(written_h[1:], S.agglomerate_hist_u8(got[:3], frags_ref, [0.3, 0.6], 75, False)):
        assert np.array_equal(open_ds(w)[:], r) and open_ds(w).attrs["bs_params"]["merge_function"] == "hist_quant_75"
    with pytest.raises(KeyError):     # not in the reference's table
        run_segmentation(str(seg_cfg), "ws", param=("merge_function=max",))


from oracle.blockwise_ref import pad_read as _pad_read, cpu_blockwise as _cpu_blockwise  # noqa: E402


def test_blockwise_segmentation_driver(tmp_path):
    """`bs segment --ws` with blockwise = true: fragments with context, per-block RAG scoring, global connected
    components, LUT and relabel, bit-equal to the oracle composition; default block = chunk shape, context = block/8."""
    import sqlite3
    from scipy.ndimage import gaussian_filter
    from bootstrapper_amd.segment import run_segmentation
    from bootstrapper_amd.zarr_io import open_ds, prepare_ds
    rng = np.random.default_rng(21)
    shape = (20, 150, 130)
    a = gaussian_filter(rng.random((3,) + shape), sigma=(0, 1, 3, 3))
    affs = ((a - a.min()) / (a.max() - a.min()) * 255).astype(np.uint8)
    affs[:, :, :40, :50] = 0                                   # an empty corner: blocks that return early
    store = str(tmp_path / "vol.zarr")
    ds = prepare_ds(store + "/affs", affs.shape, offset=(40, 8, 16), voxel_size=(40, 4, 4), chunk_shape=(3, 8, 64, 64),
                    dtype=np.uint8, axis_names=["c^", "z", "y", "x"], units=["nm"] * 3, compressor="zlib")
    ds[:] = affs
    cfg = tmp_path / "seg.toml"
    cfg.write_text(f"""affs_dataset = "{store}/affs"
fragments_dataset = "{store}/fragments"
seg_dataset_prefix = "{store}/segmentations"
blockwise = true
[db]
db_file = "{tmp_path}/rag.db"
[ws_params]
thresholds = [0.3, 0.45]
min_seed_distance = 4
filter_fragments = 0.35
remove_debris = 12
""")
    written = run_segmentation(str(cfg), "ws")
    names = [os.path.relpath(w, store) for w in written]
    assert names == ["fragments/xy--msd4--ea0--ff0.35--rd12", "segmentations/mfmean--t0.3--xy--msd4--ea0--ff0.35--rd12",
                     "segmentations/mfmean--t0.45--xy--msd4--ea0--ff0.35--rd12"]
    frags_ref, nodes, E, Sc, segs_ref = _cpu_blockwise(affs, (8, 64, 64), (1, 8, 8), 4, 0.35, 12, [0.3, 0.45])
    f = open_ds(written[0])
    assert f.dtype == np.uint64 and f.chunks == (8, 64, 64) and f.offset == (40, 8, 16)
    assert f.attrs["bs_params"]["blockwise"] is True and f.attrs["bs_params"]["filter_fragments"] == 0.35
    got = f[:]
    assert np.array_equal(got, frags_ref)
    assert len(nodes) > 100 and len(np.unique(((nodes - 1) // (8 * 64 * 64)))) > 10     # many blocks own fragments
    for w, r in zip(written[1:], segs_ref):
        seg = open_ds(w)[:]
        assert np.array_equal(seg, r)
        assert len(np.unique(seg)) < len(nodes)                                         # something merged
    lut = np.load(os.path.join(store, "luts", names[1].split("/")[1] + ".npz"))["fragment_segment_lut"]
    assert np.array_equal(lut[0], nodes)
    assert os.path.exists(os.path.join(store, "luts", names[1].split("/")[1] + ".json"))
    con = sqlite3.connect(str(tmp_path / "rag.db"))
    rows = con.execute("SELECT id, z, y, x, size FROM nodes ORDER BY id").fetchall()
    assert [r[0] for r in rows] == nodes.tolist()
    k = len(rows) // 2
    m = frags_ref == rows[k][0]
    idx = np.argwhere(m)
    assert rows[k][4] == int(m.sum())
    assert np.allclose(rows[k][1:4], np.array([40, 8, 16]) + idx.mean(axis=0) * np.array([40, 4, 4]))
    n_edges = con.execute("SELECT COUNT(*) FROM edges").fetchone()[0]
    n_null = con.execute("SELECT COUNT(*) FROM edges WHERE merge_score IS NULL").fetchone()[0]
    assert n_edges == len(E) and n_null == int(np.isnan(Sc).sum())
    con.close()

    # two workers (two slabs of block layers, fragment margins exchanged, edges gathered on rank 0): the same datasets
    cfg3 = tmp_path / "seg3.toml"
    cfg3.write_text(cfg.read_text().replace("blockwise = true", "blockwise = true\nnum_workers = 2")
                    .replace("fragments\"", "fragments_w2\"").replace("segmentations\"", "segmentations_w2\"").replace("rag.db", "rag_w2.db"))
    written3 = run_segmentation(str(cfg3), "ws")
    for a, b in zip(written, written3):
        assert np.array_equal(open_ds(a)[:], open_ds(b)[:]), (a, b)
    con = sqlite3.connect(str(tmp_path / "rag_w2.db"))
    assert con.execute("SELECT COUNT(*) FROM edges").fetchone()[0] == len(E)
    assert [r[0] for r in con.execute("SELECT id FROM nodes ORDER BY id").fetchall()] == nodes.tolist()
    con.close()

    # five workers for 3 layers x 3 rows of blocks: a 2 x 2 grid of boxes is started (a worker owns whole blocks; the fifth would
    # have none), cut along z and y -- the reference's daisy server works with any worker count, so must this
    from bootstrapper_amd.post.watershed import worker_grid
    from bootstrapper_amd.segment import get_seg_config
    cfg5 = tmp_path / "seg5.toml"
    cfg5.write_text(cfg.read_text().replace("blockwise = true", "blockwise = true\nnum_workers = 5")
                    .replace("fragments\"", "fragments_w5\"").replace("segmentations\"", "segmentations_w5\"").replace("rag.db", "rag_w5.db"))
    assert worker_grid(get_seg_config(str(cfg5), "ws")) == (2, 2)
    written5 = run_segmentation(str(cfg5), "ws")
    for a, b in zip(written, written5):
        assert np.array_equal(open_ds(a)[:], open_ds(b)[:]), (a, b)

    # a device too small for the volume (missing 4; the reference streams any volume block by block,
    # post/watershed.py:75-203): `hbm_budget_gb` between what two and what three layers of blocks need -> passes of ONE layer
    # (each holding the layer above it for context), fragments through the store, one global stitch, relabel pass by pass:
    # the datasets, the LUTs and the database of the resident run, bit for bit
    from bootstrapper_amd.volume import SlabSegmenter
    need = [SlabSegmenter.hbm_bytes((min(nl * 8, 20), 150, 130), (1, 8, 8), 2, nl * 9) for nl in (2, 3)]
    cfg6 = tmp_path / "seg6.toml"
    cfg6.write_text(cfg.read_text().replace("blockwise = true", f"blockwise = true\nhbm_budget_gb = {(need[0] + need[1]) / 2 / 2**30:.6f}")
                    .replace("fragments\"", "fragments_st\"").replace("segmentations\"", "segmentations_st\"").replace("rag.db", "rag_st.db"))
    import bootstrapper_amd.post.watershed as W
    calls = []
    orig = W._waterz_streamed
    W._waterz_streamed = lambda *a, **k: (calls.append(a[-1]), orig(*a, **k))[1]
    try:
        written6 = run_segmentation(str(cfg6), "ws")
    finally:
        W._waterz_streamed = orig
    assert calls == [1]                                           # one layer per pass
    for a, b in zip(written, written6):
        assert np.array_equal(open_ds(a)[:], open_ds(b)[:]), (a, b)
    lut6 = np.load(os.path.join(store, "luts_st", os.path.basename(written6[1]) + ".npz"))["fragment_segment_lut"]
    assert np.array_equal(lut6, lut)
    con = sqlite3.connect(str(tmp_path / "rag_st.db"))
    assert con.execute("SELECT COUNT(*) FROM edges").fetchone()[0] == len(E)
    assert [r[0] for r in con.execute("SELECT id FROM nodes ORDER BY id").fetchall()] == nodes.tolist()
    assert con.execute("SELECT COUNT(*) FROM edges WHERE merge_score IS NULL").fetchone()[0] == int(np.isnan(Sc).sum())
    con.close()
    with pytest.raises(MemoryError):                              # not even two layers
        cfg7 = tmp_path / "seg7.toml"
        cfg7.write_text(cfg6.read_text().replace(f"hbm_budget_gb = {(need[0] + need[1]) / 2 / 2**30:.6f}", "hbm_budget_gb = 0.001"))
        run_segmentation(str(cfg7), "ws")

    # block_shape = "roi": one block, no context (post/watershed.py:357-363), same machinery
    cfg2 = tmp_path / "seg2.toml"
    cfg2.write_text(cfg.read_text().replace("blockwise = true", 'blockwise = true\nblock_shape = "roi"')
                    .replace("fragments\"", "fragments_roi\"").replace("segmentations\"", "segmentations_roi\""))
    written2 = run_segmentation(str(cfg2), "ws")
    fr2, _, _, _, segs2 = _cpu_blockwise(affs, shape, (0, 0, 0), 4, 0.35, 12, [0.3, 0.45])
    assert open_ds(written2[0]).attrs["bs_params"]["blockwise"] is False
    assert np.array_equal(open_ds(written2[0])[:], fr2)
    assert np.array_equal(open_ds(written2[2])[:], segs2[1])


def test_cc_segmentation_driver(tmp_path):
    """`bs segment --cc`: dataset names, attributes and contents of the thresholded-affinity connected components."""
    from scipy.ndimage import gaussian_filter
    from bootstrapper_amd.segment import run_segmentation
    from bootstrapper_amd.zarr_io import open_ds, prepare_ds
    from oracle import seg_ref as S
    rng = np.random.default_rng(2)
    a = gaussian_filter(rng.random((3, 10, 60, 50)), sigma=(0, 1, 2, 2))
    affs = ((a - a.min()) / (a.max() - a.min()) * 255).astype(np.uint8)
    store = str(tmp_path / "v.zarr")
    ds = prepare_ds(store + "/affs", affs.shape, offset=(0, 0, 0), voxel_size=(40, 4, 4), chunk_shape=(3, 8, 32, 32), dtype=np.uint8,
                    axis_names=["c^", "z", "y", "x"], units=["nm"] * 3)
    ds[:] = affs
    cfg = tmp_path / "seg.toml"
    cfg.write_text(f'affs_dataset = "{store}/affs"\nfragments_dataset = "{store}/fragments"\nseg_dataset_prefix = "{store}/segmentations"\n'
                   'blockwise = false\n[cc_params]\nthreshold = 0.55\nremove_debris = 20\n')
    written = run_segmentation(str(cfg), "cc")
    assert [os.path.relpath(w, store) for w in written] == ["fragments/t0.55", "segmentations/t0.55--rd20"]
    ref, _ = S.cc_affs_u8(affs, 0.55)
    f = open_ds(written[0])
    assert f.attrs["bs_params"]["method"] == "cc" and np.array_equal(f[:], ref.astype(np.uint64))
    counts = np.bincount(ref.ravel())
    keep = counts >= 20
    keep[0] = False
    assert np.array_equal(open_ds(written[1])[:], np.where(keep[ref], ref, 0).astype(np.uint64))


def test_refine_filters(tmp_path):
    """`bs refine` size / outlier / z filters and remap against a numpy restatement of the reference's formulas
    (refine.py:98-307; fastremap and daisy are absent, so the reference file itself cannot run here)."""
    from bootstrapper_amd import refine as R
    from bootstrapper_amd.zarr_io import open_ds, prepare_ds
    rng = np.random.default_rng(6)
    vol = np.zeros((24, 90, 70), dtype=np.uint64)
    nid = 1
    for _ in range(60):
        z, y, x = rng.integers(0, 20), rng.integers(0, 80), rng.integers(0, 60)
        dz, dy, dx = rng.integers(1, 8), rng.integers(2, 25), rng.integers(2, 25)
        vol[z:z + dz, y:y + dy, x:x + dx] = nid * 1_000_003 + (1 << 40)      # sparse 64-bit ids
        nid += 1
    store = str(tmp_path / "s.zarr")
    ds = prepare_ds(store + "/seg", vol.shape, offset=(0, 0, 0), voxel_size=(40, 4, 4), chunk_shape=(8, 32, 32), dtype=np.uint64,
                    axis_names=["z", "y", "x"], units=["nm"] * 3, compressor="zlib")
    ds[:] = vol
    ids, sizes = np.unique(vol[vol > 0], return_counts=True)
    zz = np.nonzero(vol)
    zmin = {i: int(zz[0][vol[zz] == i].min()) for i in ids}
    zmax = {i: int(zz[0][vol[zz] == i].max()) for i in ids}
    tid, tsz, tzmin, tzmax = R.label_table(open_ds(store + "/seg"))
    assert np.array_equal(tid, ids) and np.array_equal(tsz, sizes)
    assert [zmin[i] for i in ids] == tzmin.tolist() and [zmax[i] for i in ids] == tzmax.tolist()

    def masked(remove):
        out = vol.copy()
        out[np.isin(out, remove)] = 0
        return out
    out = R.size_filter(store + "/seg", min_size=300, max_size=4000)
    assert out == store + "/seg_size_filtered"
    o = open_ds(out)
    assert o.chunks == (8, 32, 32) and o.voxel_size == (40, 4, 4)
    assert np.array_equal(o[:], masked(ids[(sizes < 300) | (sizes > 4000)]))
    stat = sizes[sizes >= 50]
    lo, hi = stat.mean() - 1.0 * stat.std(), stat.mean() + 1.0 * stat.std()
    out = R.outlier_filter(store + "/seg", num_std=1.0, min_size=50)
    assert np.array_equal(open_ds(out)[:], masked(ids[(sizes < lo) | (sizes > hi)]))
    out = R.z_filter(store + "/seg", min_z=3)
    assert np.array_equal(open_ds(out)[:], masked(np.array([i for i in ids if zmax[i] - zmin[i] + 1 <= 3], dtype=np.uint64)))
    a, b, c = int(ids[3]), int(ids[7]), int(ids[11])
    out = R.remap(store + "/seg", remove_ids=f"{a}", merge_ids=(f"{b},{c}",))
    want = vol.copy()
    want[want == a] = 0
    want[want == c] = b
    assert out == store + "/seg_remapped" and np.array_equal(open_ds(out)[:], want)
    assert R.size_filter(store + "/seg", min_size=10, dry_run=True) is None


@pytest.mark.parametrize("precision", [None, "f32"])
def test_bootstrap_chain_2d_mtlsd_then_second_stage(tmp_path, golden_dir, precision):
    """The chain of the CREMI example (2d_mtlsd -> 3d_affs_from_2d_mtlsd) through `run_prediction`: a 2-D setup
    predicted as stacks of sections, then a second-stage setup reading both prediction datasets; each compared
    with the oracle applied block by block to the same data.  precision None = the drivers' default (split-bf16:
    2-D setups, (1,3,3) kernels, u8/255 inputs and num_fmaps_out in that arithmetic), "f32" = exact f32 products."""
    kw = {} if precision is None else {"precision": precision}
    from bootstrapper_amd.predict import run_prediction
    from bootstrapper_amd.zarr_io import open_ds, prepare_ds
    from oracle import unet_ref as R
    from test_oracle_unet import family_case
    nc1, sd1, _, _, _ = family_case(golden_dir, "2d_mtlsd_f4i2")
    nc2, sd2, _, _, _ = family_case(golden_dir, "from_2d_mtlsd_f3i2")
    # block shapes for this small volume: 2-D net 108 -> 16 (+8), second stage (22,100,100) -> (2,8,8) (+2,+8,+8)
    nc1.update(input_shape=[108, 108], output_shape=[16, 16], shape_increase=[8, 8])
    nc2.update(input_shape=[22, 100, 100], output_shape=[2, 8, 8], shape_increase=[2, 8, 8])
    store = str(tmp_path / "vol.zarr")
    rng = np.random.default_rng(5)
    raw = rng.integers(0, 256, size=(19, 40, 45), dtype=np.uint8)
    ds = prepare_ds(store + "/raw", raw.shape, offset=(0, 0, 0), voxel_size=(40, 4, 4), chunk_shape=(8, 32, 32),
                    dtype=np.uint8, axis_names=["z", "y", "x"], units=["nm"] * 3)
    ds[:] = raw
    toml = []
    for i, (name, nc, sd) in enumerate((("2d_mtlsd", nc1, sd1), ("3d_affs_from_2d_mtlsd", nc2, sd2))):
        setup = tmp_path / name
        setup.mkdir()
        (setup / "net_config.json").write_text(json.dumps(nc))
        torch.save({"model_state_dict": {k: torch.from_numpy(v) for k, v in sd.items()}}, str(setup / "model_checkpoint_5.ckpt"))
        ins = ([f"{store}/raw"] if i == 0 else [f"{store}/predictions/5/2d_lsds", f"{store}/predictions/5/2d_affs"])
        toml.append(f'["0{i + 1}-{name}"]\nsetup_dir = "{setup}"\ninput_datasets = {json.dumps(ins)}\n'
                    f'checkpoint = "{setup}/model_checkpoint_5"\noutput_datasets_prefix = "{store}/predictions"\n'
                    f'chain_str = "{"" if i == 0 else "2d_mtlsd"}"\nnum_workers = 1\nnum_gpus = 1\n')
    cfg = tmp_path / "pred.toml"
    cfg.write_text("\n".join(toml))
    run_prediction(str(cfg), "01", **kw)
    lsds, affs = open_ds(store + "/predictions/5/2d_lsds"), open_ds(store + "/predictions/5/2d_affs")
    assert lsds.shape == (6, 19, 40, 45) and lsds.chunks == (6, 1, 24, 24) and affs.dtype == np.uint8
    got = [lsds[:], affs[:]]
    # oracle, section by section like the reference worker: 3 adjacent sections (reflect padded) -> 1 section
    full = np.pad(raw, [(1, 1), (46, 46 + 24), (46, 46 + 24)], mode="reflect")
    ref = [np.zeros_like(g) for g in got]
    for z in range(19):
        for y in range(0, 40, 24):
            for x in range(0, 45, 24):
                blk = full[z:z + 3, y:y + 116, x:x + 116]
                outs = R.family_forward(nc1, sd1, torch.from_numpy(R.normalize_raw(blk)[None, :, None]))
                hy, hx = min(24, 40 - y), min(24, 45 - x)
                for r, o in zip(ref, outs):
                    r[:, z, y:y + hy, x:x + hx] = R.to_u8(o.numpy())[:, 0, :hy, :hx]
    for g, r in zip(got, ref):
        diff = np.abs(g.astype(np.int32) - r.astype(np.int32))
        assert diff.max() <= 1 and (diff == 0).mean() > 0.99

    run_prediction(str(cfg), "02", **kw)
    out = open_ds(store + "/predictions/5--from--2d_mtlsd/3d_affs")
    assert out.shape == (9, 19, 40, 45) and out.chunks == (9, 4, 16, 16)
    got2 = out[:]
    both = np.concatenate(got, axis=0)                                   # (12, D, H, W) u8, lsds then affs
    full = np.pad(both, [(0, 0), (10, 10 + 4), (46, 46 + 16), (46, 46 + 16)], mode="reflect")
    ref2 = np.zeros_like(got2)
    for z in range(0, 19, 4):
        for y in range(0, 40, 16):
            for x in range(0, 45, 16):
                blk = full[:, z:z + 24, y:y + 108, x:x + 108]
                o = R.to_u8(R.family_forward(nc2, sd2, torch.from_numpy(R.normalize_unit(blk)[None]))[0].numpy())
                hz, hy, hx = min(4, 19 - z), min(16, 40 - y), min(16, 45 - x)
                ref2[:, z:z + hz, y:y + hy, x:x + hx] = o[:, :hz, :hy, :hx]
    diff = np.abs(got2.astype(np.int32) - ref2.astype(np.int32))
    assert diff.max() <= 1 and (diff == 0).mean() > 0.99


def test_predict_two_workers_failure_injection_and_worker_script(tmp_path, golden_dir, monkeypatch):
    """`run_prediction` with num_gpus = 2 (two worker processes; on a one-GPU box they share the card), each reading only
    the slab of the input its blocks need: bit-equal to the one-worker run.  Then the failure path: a block that raises
    twice is retried and the output is unchanged; a block that always raises ends in the reference's RuntimeError
    (blockwise.py:12-22).  Then the per-setup worker script contract (models/3d_affs/predict.py:19-58)."""
    import subprocess
    import sys
    from bootstrapper_amd import predict as P
    from bootstrapper_amd.unet import Model
    from bootstrapper_amd.zarr_io import open_ds
    nc, sd, raw, store, cfg = _setup(tmp_path, golden_dir)
    P.run_prediction(cfg, "01", precision="f32")
    one = open_ds(store + "/predictions/1000/3d_affs")[:]
    assert one.std() > 5
    cfg2 = str(tmp_path / "pred2.toml")
    open(cfg2, "w").write(open(cfg).read().replace("num_gpus = 1", "num_gpus = 2").replace("/predictions", "/predictions2"))
    P.run_prediction(cfg2, "01", precision="f32")
    assert np.array_equal(open_ds(store + "/predictions2/1000/3d_affs")[:], one)

    # failure injection (one worker, in process)
    real = Model.predict_u8
    calls = {"n": 0}

    def flaky(self, raw_u8, want_f32=False):
        calls["n"] += 1
        if calls["n"] in (3, 4):                                   # the third block fails twice, then goes through
            raise RuntimeError("injected")
        return real(self, raw_u8, want_f32)
    monkeypatch.setattr(Model, "predict_u8", flaky)
    cfg3 = str(tmp_path / "pred3.toml")
    open(cfg3, "w").write(open(cfg).read().replace("/predictions", "/predictions3"))
    P.run_prediction(cfg3, "01", precision="f32")
    assert np.array_equal(open_ds(store + "/predictions3/1000/3d_affs")[:], one)
    n_blocks = len(P.enumerate_blocks(P.get_pred_config(cfg3, "01-3d_affs")))
    assert calls["n"] == n_blocks + 2

    state = {"k": 0}

    def broken(self, raw_u8, want_f32=False):
        state["k"] += 1
        if 2 < state["k"] <= 2 + 6:                                 # all six attempts of the third block
            raise RuntimeError("injected")
        return real(self, raw_u8, want_f32)
    monkeypatch.setattr(Model, "predict_u8", broken)
    with pytest.raises(RuntimeError, match=rf"task PredictBlockwiseTask: 1 failed, 0 orphaned of {n_blocks} blocks"):
        P.run_prediction(cfg3, "01", precision="f32")
    monkeypatch.setattr(Model, "predict_u8", real)

    # the worker script: argv of the reference, datasets prepared by the caller, ROI given in world units
    pc = P.get_pred_config(cfg, "01-3d_affs")
    pc["output_datasets"] = [store + "/by_worker/3d_affs"]
    P.prepare_outputs(pc, open_ds(store + "/raw"))
    setup_dir = pc["setup_dir"]
    r = subprocess.run([sys.executable, "-m", "bootstrapper_amd.predict", "-c", pc["checkpoint"], "-i", store + "/raw", "-o",
                        store + "/by_worker/3d_affs", "-ro", "80 8 8", "-rs", f"{23 * 40} {50 * 4} {61 * 4}", "-n", "1",
                        "--setup-dir", setup_dir, "--precision", "f32"],
                       capture_output=True, text=True, timeout=600, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert np.array_equal(open_ds(store + "/by_worker/3d_affs")[:], one)


def test_cremi_shaped_volume_full_net_then_blockwise_segment_and_filter(tmp_path, monkeypatch):
    """BASELINE configs 2 and 5 in one chain, on a CREMI-shaped synthetic volume (125 x 1250 x 1250 voxels of (40,4,4) nm,
    examples/cremi/README.md:19-24): `run_prediction` with the full 3d_affs network in 128^3 blocks over a sub-ROI
    (a single layer of blocks that overhangs the 125 sections, and overhangs the ROI in y and x: writes are clipped),
    three blocks against the CPU oracle in the split-bf16 mode, then `bs segment --ws` blockwise on the predicted
    affinities and `bs refine` size filter on the result."""
    from bootstrapper_amd.predict import run_prediction
    from bootstrapper_amd.segment import run_segmentation
    from bootstrapper_amd import refine as RF
    from bootstrapper_amd.synth import synthetic_state_dict, synthetic_volume
    from bootstrapper_amd.zarr_io import open_ds, prepare_ds
    from oracle import unet_ref as R
    from oracle.blockwise_ref import cpu_blockwise
    from tests.test_lib_cpu import AFFS_NET_CONFIG as NC
    nc = dict(NC, input_shape=[32, 196, 196], output_shape=[4, 104, 104], shape_increase=[124, 24, 24],
              inputs={"raw": {"dims": 1}}, outputs={"3d_affs": {"dtype": "uint8", "dims": 6}})
    setup = tmp_path / "setup_03"
    setup.mkdir()
    (setup / "net_config.json").write_text(json.dumps(nc))
    sd = synthetic_state_dict(NC, 0)
    torch.save({"state_dict": {"model." + k: torch.from_numpy(v) for k, v in sd.items()}}, str(setup / "model_checkpoint_3000.ckpt"))
    raw = synthetic_volume((125, 1250, 1250), 0).cpu().numpy()
    store = str(tmp_path / "cremi.zarr")
    ds = prepare_ds(store + "/raw", raw.shape, offset=(0, 0, 0), voxel_size=(40, 4, 4), chunk_shape=(25, 250, 250), dtype=np.uint8,
                    axis_names=["z", "y", "x"], units=["nm"] * 3)
    ds[:] = raw
    cfg = tmp_path / "pred.toml"
    oy, ox, ny, nx = 300, 500, 300, 290                      # ROI in voxels: 3 x 3 blocks, the last ones overhang it
    cfg.write_text(f'["03-3d_affs"]\nsetup_dir = "{setup}"\ninput_datasets = ["{store}/raw"]\ncheckpoint = "{setup}/model_checkpoint_3000"\n'
                   f'output_datasets_prefix = "{store}/predictions"\nchain_str = ""\nnum_workers = 1\nnum_gpus = 1\n'
                   f'roi_offset = [0, {oy * 4}, {ox * 4}]\nroi_shape = [{125 * 40}, {ny * 4}, {nx * 4}]\n')
    run_prediction(str(cfg), "03")                          # default precision: bf16x3
    out = open_ds(store + "/predictions/3000/3d_affs")
    assert out.shape == (6, 125, ny, nx) and out.chunks == (6, 125, 128, 128) and out.offset == (0, oy * 4, ox * 4)
    got = out[:]
    # the same job with two predict lanes (`pred_lanes`: two engines, blocks dealt alternately, their forward passes overlapping): same bits
    cfg2 = tmp_path / "pred_lanes.toml"
    cfg2.write_text(cfg.read_text().replace(f'output_datasets_prefix = "{store}/predictions"', f'output_datasets_prefix = "{store}/predictions_lanes"')
                    + "pred_lanes = 2\n")
    monkeypatch.setenv("BSMI_PRED_LANES_WAIT", "1")   # the second engine from the first block on (else it joins when its weights are packed)
    run_prediction(str(cfg2), "03")
    monkeypatch.delenv("BSMI_PRED_LANES_WAIT")
    assert np.array_equal(open_ds(store + "/predictions_lanes/3000/3d_affs")[:], got)
    full = np.pad(raw, [(14, 14 + 128), (46, 46 + 128), (46, 46 + 128)], mode="reflect")
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    for (y, x) in ((0, 0), (128, 256), (256, 128)):          # a corner block, two blocks that overhang the ROI
        blk = full[0:156, oy + y:oy + y + 220, ox + x:ox + x + 220]
        ref = R.to_u8(R.predict_block(R.default_cfg(12, 5), sd, blk, ["affs_head"])[0])
        hy, hx = min(128, ny - y), min(128, nx - x)
        diff = np.abs(got[:, :, y:y + hy, x:x + hx].astype(np.int32) - ref[:, :125, :hy, :hx].astype(np.int32))
        assert diff.max() <= 1 and (diff == 0).mean() > 0.995, (y, x, diff.max())
    # blockwise segmentation of the predicted affinities (block = chunk = 128^3 clipped to 125 sections, context 16)
    seg_cfg = tmp_path / "seg.toml"
    seg_cfg.write_text(f'affs_dataset = "{store}/predictions/3000/3d_affs"\nfragments_dataset = "{store}/fragments"\n'
                       f'seg_dataset_prefix = "{store}/segmentations"\nblockwise = true\n[db]\ndb_file = "{tmp_path}/rag.db"\n'
                       f'[ws_params]\nthresholds = [0.35]\n')
    written = run_segmentation(str(seg_cfg), "ws")
    frags_ref, nodes, _, _, segs_ref = cpu_blockwise(got[:3], (125, 128, 128), (15, 16, 16), 10, 0.1, 64, [0.35], workers=8)  # segment.py:16 defaults
    assert np.array_equal(open_ds(written[0])[:], frags_ref)
    seg = open_ds(written[1])[:]
    assert np.array_equal(seg, segs_ref[0]) and len(np.unique(seg)) < len(nodes)
    # bs refine size filter on the segmentation dataset
    ids, counts = np.unique(seg[seg > 0], return_counts=True)
    out_ds = RF.size_filter(written[1], min_size=500, max_size=10 ** 9)
    want = seg.copy()
    want[np.isin(want, ids[counts < 500])] = 0
    assert np.array_equal(open_ds(out_ds)[:], want) and (want > 0).any()
