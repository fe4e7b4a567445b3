"""`bs refine` statistics filters and id remap on the device.

Behavioural mirror of /root/reference/bootstrapper/refine.py: `size_filter` (:176-213), `outlier_filter` (:131-168),
`z_filter` (:221-257), `remap` (:272-307) with the same options, statistics formulas, default output names
(`_default_out`, :24-31) and dataset metadata (`_prepare_like`, :34-45).  The label scans (`_global_sizes`, z extents)
and the masking / remapping run in libbsmi (`bsmi_label_table_u64`, `bsmi_lut_relabel`), tile by tile; the decisions
on the per-object table are a few numpy lines, as in the reference.  `morph` (fastmorph [EXT]) is not part of this engine.
"""
import click
import numpy as np

from .zarr_io import open_ds, prepare_ds


def _default_out(in_array, suffix):
    head, sep, tail = in_array.partition(".zarr/")
    if not sep or ".zarr/" in tail:
        raise click.ClickException(f"cannot derive an out_array from {in_array!r}; pass --out_array")
    return f"{head}.zarr/{tail}_{suffix}"


def _prepare_like(in_ds, out_array):
    return prepare_ds(out_array, shape=in_ds.shape, offset=in_ds.offset, voxel_size=in_ds.voxel_size, axis_names=in_ds.axis_names,
                      units=in_ds.units, dtype=in_ds.dtype, chunk_shape=in_ds.chunks, compressor=in_ds.meta.get("compressor"))


def _tiles(in_ds, tile=1024):
    nz, ny, nx = in_ds.shape
    cz = int(in_ds.chunks[0])
    for iz in range(0, nz, cz):
        for iy in range(0, ny, tile):
            for ix in range(0, nx, tile):
                yield iz, (slice(iz, min(iz + cz, nz)), slice(iy, min(iy + tile, ny)), slice(ix, min(ix + tile, nx)))


class _Device:
    def __init__(self, in_ds, device=0, tile=1024):
        import torch
        from .post.engine import SegEngine
        self.torch = torch
        self.dev = torch.device("cuda", device)
        cz = int(in_ds.chunks[0])
        self.engine = SegEngine((min(cz, in_ds.shape[0]), min(tile, in_ds.shape[1]), min(tile, in_ds.shape[2])), device)
        self.tile = tile

    def upload(self, block):
        return self.torch.from_numpy(np.ascontiguousarray(block).astype(np.uint64).view(np.int64)).to(self.dev)


def label_table(in_ds, device=0):
    """-> (ids ascending u64, sizes, zmin, zmax) over the whole volume (refine.py:98-109, 228-250)."""
    d = _Device(in_ds, device)
    parts = []
    for iz, sl in _tiles(in_ds, d.tile):
        parts.append(d.engine.label_table(d.upload(in_ds[sl]), iz))
    if not parts:
        z = np.zeros(0, np.int64)
        return np.zeros(0, np.uint64), z, z, z
    ids = np.concatenate([p[0] for p in parts])
    uniq, inv = np.unique(ids, return_inverse=True)
    sizes = np.bincount(inv, weights=np.concatenate([p[1] for p in parts]).astype(np.float64)).astype(np.int64)
    zmin = np.full(uniq.size, np.iinfo(np.int64).max)
    zmax = np.full(uniq.size, np.iinfo(np.int64).min)
    np.minimum.at(zmin, inv, np.concatenate([p[2] for p in parts]))
    np.maximum.at(zmax, inv, np.concatenate([p[3] for p in parts]))
    return uniq, sizes, zmin, zmax


def _apply_mapping(in_ds, out_array, keys, vals, device=0):
    """out = in with keys[k] -> vals[k] (ids not listed stay), tile by tile through bsmi_lut_relabel."""
    from .post.engine import lut_relabel
    d = _Device(in_ds, device)
    out_ds = _prepare_like(in_ds, out_array)
    order = np.argsort(keys, kind="stable")
    k = d.torch.from_numpy(np.asarray(keys, np.uint64)[order].view(np.int64)).to(d.dev)
    v = d.torch.from_numpy(np.asarray(vals, np.uint64)[order].view(np.int64)).to(d.dev)
    for _, sl in _tiles(in_ds, d.tile):
        lab = d.upload(in_ds[sl])
        out_ds[sl] = lut_relabel(lab, k, v).cpu().numpy().view(np.uint64).astype(in_ds.dtype)
    return out_ds


def _finish_filter(in_ds, in_array, out_array, remove_ids, dry_run, suffix, device=0):
    if dry_run:
        print("dry run; nothing written")
        return None
    out_array = out_array or _default_out(in_array, suffix)
    print(f"Writing to {out_array}")
    _apply_mapping(in_ds, out_array, remove_ids, np.zeros(len(remove_ids), np.uint64), device)
    return out_array


def size_filter(in_array, out_array=None, min_size=0, max_size=None, dry_run=False, device=0):
    in_ds = open_ds(in_array)
    uniq, sizes, _, _ = label_table(in_ds, device)
    if uniq.size == 0:
        raise click.ClickException("no foreground objects in volume")
    remove = np.zeros(uniq.size, dtype=bool)
    if min_size > 0:
        remove |= sizes < min_size
    if max_size:
        remove |= sizes > max_size
    remove_ids = uniq[remove]
    print(f"{uniq.size} objects; sizes min={int(sizes.min())} max={int(sizes.max())} median={int(np.median(sizes))}")
    print(f"range [{min_size}, {max_size}] -> removing {remove_ids.size} objects")
    return _finish_filter(in_ds, in_array, out_array, remove_ids, dry_run, "size_filtered", device)


def outlier_filter(in_array, out_array=None, num_std=3.0, min_size=0, dry_run=False, device=0):
    in_ds = open_ds(in_array)
    uniq, sizes, _, _ = label_table(in_ds, device)
    if uniq.size == 0:
        raise click.ClickException("no foreground objects in volume")
    stat_sizes = sizes[sizes >= min_size]
    if stat_sizes.size == 0:
        raise click.ClickException(f"no objects with size >= min_size ({min_size})")
    mean, std = float(stat_sizes.mean()), float(stat_sizes.std())
    lo, hi = mean - num_std * std, mean + num_std * std
    remove_ids = uniq[(sizes < lo) | (sizes > hi)]
    p50, p90, p99, p999 = np.percentile(stat_sizes, [50, 90, 99, 99.9])
    print(f"{uniq.size} objects; {stat_sizes.size} with size >= {min_size}")
    print(f"size p50={p50:.0f} p90={p90:.0f} p99={p99:.0f} p99.9={p999:.0f} min={int(sizes.min())} max={int(sizes.max())}")
    print(f"mean={mean:.1f} std={std:.1f} | cut lo={lo:.1f} hi={hi:.1f} -> removing {remove_ids.size} objects")
    return _finish_filter(in_ds, in_array, out_array, remove_ids, dry_run, "outlier_filtered", device)


def z_filter(in_array, out_array=None, min_z=1, dry_run=False, device=0):
    in_ds = open_ds(in_array)
    ids, _, zmin, zmax = label_table(in_ds, device)
    spans = zmax - zmin + 1
    remove_ids = ids[spans <= min_z]
    print(f"{ids.size} objects; removing {remove_ids.size} with z-extent <= {min_z}")
    return _finish_filter(in_ds, in_array, out_array, remove_ids, dry_run, "z_filtered", device)


def remap(in_array, out_array=None, remove_ids=None, merge_ids=(), device=0):
    remove = {int(x) for x in remove_ids.replace(" ", "").split(",")} if remove_ids else set()
    merge = {}
    for group in merge_ids:
        ids = [int(x) for x in group.replace(" ", "").split(",")]
        for mid in ids:
            merge[mid] = ids[0]
    conflict = remove & set(merge)
    if conflict:
        raise click.ClickException(f"ids given to both --remove_ids and --merge_ids: {sorted(conflict)}")
    mapping = {**{i: 0 for i in remove}, **merge}
    if not mapping:
        raise click.ClickException("nothing to do: pass --remove_ids and/or --merge_ids")
    print(f"remapping {len(mapping)} ids: {mapping}")
    in_ds = open_ds(in_array)
    out_array = out_array or _default_out(in_array, "remapped")
    print(f"Writing to {out_array}")
    keys = np.array(sorted(mapping), dtype=np.uint64)
    _apply_mapping(in_ds, out_array, keys, np.array([mapping[int(k)] for k in keys], dtype=np.uint64), device)
    return out_array


@click.group()
def refine():
    """Refine segmented volumes: size/outlier/z filtering, id remap."""


def _cmd(name, fn, options):
    f = lambda **kw: fn(**kw)  # noqa: E731
    f.__doc__ = fn.__doc__ or name
    for opt in reversed(options):
        f = opt(f)
    return refine.command(name)(f)


_io = [click.option("--in_array", "-i", type=click.Path(exists=True), required=True), click.option("--out_array", "-o", type=click.Path())]
_dry = click.option("--dry_run", is_flag=True, default=False)
_cmd("size_filter", size_filter, _io + [click.option("--min_size", type=int, default=0), click.option("--max_size", type=int, default=None), _dry])
_cmd("outlier_filter", outlier_filter, _io + [click.option("--num_std", "-n", type=float, default=3.0), click.option("--min_size", type=int, default=0), _dry])
_cmd("z_filter", z_filter, _io + [click.option("--min_z", "-z", type=int, default=1), _dry])
_cmd("remap", remap, _io + [click.option("--remove_ids", "-r", type=str, default=None), click.option("--merge_ids", "-m", type=str, multiple=True)])
