"""Host-side drivers: Zarr I/O, predict block geometry and dataset naming (CPU only)."""
import json
import os

import numpy as np
import pytest


def test_zarr_roundtrip_and_attrs(tmp_path):
    from bootstrapper_amd.zarr_io import prepare_ds, open_ds
    p = str(tmp_path / "v.zarr" / "volumes" / "raw")
    a = prepare_ds(p, (20, 33, 41), offset=(40, 8, 8), voxel_size=(40, 4, 4), chunk_shape=(8, 16, 16),
                   dtype=np.uint8, units=["nm"] * 3, axis_names=["z", "y", "x"])
    x = np.random.default_rng(0).integers(0, 256, (20, 33, 41), dtype=np.uint8)
    a[:] = x
    b = open_ds(p)
    assert np.array_equal(b[:], x)
    assert np.array_equal(b[3:17, 5:30, 7:40], x[3:17, 5:30, 7:40])
    a[2:5, 10:20, 3:9] = 7
    x[2:5, 10:20, 3:9] = 7
    assert np.array_equal(open_ds(p)[:], x)
    assert b.roi == ((40, 8, 8), (800, 132, 164)) and b.voxel_size == (40, 4, 4)
    assert b.roi_to_slices((80, 16, 16), (400, 64, 64)) == (slice(1, 11), slice(2, 18), slice(2, 18))
    with pytest.raises(ValueError):
        b.roi_to_slices((81, 16, 16), (400, 64, 64))
    with pytest.raises(PermissionError):
        b[0:1] = 0
    for comp in ("zlib", "gzip", "zstd", "lz4", "blosc", "default", None,
                 {"id": "blosc", "cname": "zstd", "clevel": 3, "shuffle": 2, "blocksize": 0}):
        q = str(tmp_path / "v.zarr" / f"c{comp if not isinstance(comp, dict) else 'dict'}")
        c = prepare_ds(q, (3, 9, 9), chunk_shape=(3, 4, 4), dtype=np.uint64, compressor=comp)
        y = np.arange(243, dtype=np.uint64).reshape(3, 9, 9)
        c[:] = y
        d = open_ds(q)
        assert np.array_equal(d[:], y)
        assert np.array_equal(d[1:3, 2:7, 3:9], y[1:3, 2:7, 3:9])
    # new arrays carry zarr-python's default compressor, like the reference's prepare_ds
    assert json.load(open(os.path.join(p, ".zarray")))["compressor"] == \
        {"id": "blosc", "cname": "lz4", "clevel": 5, "shuffle": 1, "blocksize": 0}
    # unsupported codec is a clear error, not garbage
    meta = json.load(open(os.path.join(p, ".zarray")))
    meta["compressor"] = {"id": "bz2", "level": 1}
    json.dump(meta, open(os.path.join(p, ".zarray"), "w"))
    with pytest.raises(NotImplementedError):
        open_ds(p)


def test_zarr_reads_chunks_written_by_c_blosc(tmp_path):
    """An array whose chunk files are frames produced by c-blosc 1.21.0 (tests/golden/codec_cases.npz), as
    zarr-python would have written them."""
    from bootstrapper_amd.zarr_io import open_ds
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "codec_cases.npz"))
    want = g["plain__raw_u8"].reshape(8, 40, 40)
    for cname, shuffle in (("lz4", 1), ("zstd", 2), ("blosclz", 0), ("lz4hc", 1), ("zlib", 1)):
        path = tmp_path / "ext.zarr" / f"{cname}{shuffle}"
        os.makedirs(path)
        json.dump({"zarr_format": 2, "shape": [16, 40, 40], "chunks": [8, 40, 40], "dtype": "|u1", "fill_value": 0, "order": "C",
                   "filters": None, "compressor": {"id": "blosc", "cname": cname, "clevel": 5, "shuffle": shuffle, "blocksize": 0}},
                  open(path / ".zarray", "w"))
        json.dump({"resolution": [40, 4, 4], "offset": [0, 0, 0]}, open(path / ".zattrs", "w"))
        with open(path / "1.0.0", "wb") as f:
            f.write(g[f"blosc__raw_u8__{cname}__s{shuffle}__b0"].tobytes())
        a = open_ds(str(path))
        assert a.voxel_size == (40, 4, 4)
        got = a[:]
        assert np.array_equal(got[8:], want) and not got[:8].any()      # chunk 0.0.0 is missing: fill_value
        assert np.array_equal(a[10:13, 5:30, 7:40], want[2:5, 5:30, 7:40])


def test_block_geometry_and_names():
    """reference predict.py:114-131,143-155; halo of the 3-D nets = (14,46,46) voxels."""
    from bootstrapper_amd.predict import block_rois, output_dataset_names, enumerate_blocks
    nc = {"shape_increase": [0, 216, 216], "input_shape": [32, 196, 196], "output_shape": [4, 104, 104]}
    r = block_rois(nc, (40, 4, 4))
    assert r["input_shape"] == [32, 412, 412] and r["output_shape"] == [4, 320, 320]
    assert r["context"] == [14 * 40, 46 * 4, 46 * 4]
    assert r["read_roi"] == ([-560, -184, -184], [1280, 1648, 1648]) and r["write_roi"] == ([0, 0, 0], [160, 1280, 1280])
    assert output_dataset_names("/s/model_checkpoint_30000", "/d/v.zarr/predictions", {"outputs": {"3d_affs": {}}}) == \
        ["/d/v.zarr/predictions/30000/3d_affs"]
    assert output_dataset_names("/s/model_checkpoint_5000", "p", {"outputs": {"3d_lsds": {}, "3d_affs": {}}}, "2d_lsd") == \
        ["p/5000--from--2d_lsd/3d_lsds", "p/5000--from--2d_lsd/3d_affs"]
    # CREMI-sized volume (125,1250,1250) in default blocks: overhang fit
    cfg = {"output_roi": ([0, 0, 0], [125 * 40, 1250 * 4, 1250 * 4]), "voxel_size": [40, 4, 4], "output_shape": [4, 320, 320]}
    blocks = enumerate_blocks(cfg)
    assert len(blocks) == 32 * 4 * 4 and blocks[0] == (0, 0, 0) and blocks[-1] == (124, 960, 960)


def test_cli_has_reference_commands():
    from click.testing import CliRunner
    from bootstrapper_amd.cli import cli
    r = CliRunner().invoke(cli, ["--help"])
    assert r.exit_code == 0
    for name in ("predict", "segment", "p", "s"):
        assert name in r.output
    r = CliRunner().invoke(cli, ["segment", "--help"])
    for flag in ("-ws", "-mws", "-cc", "-ro", "-rs", "-b", "-n", "-bs", "-bc", "-p"):
        assert flag in r.output
