"""`bs refine` statistics filters and id remap on the device.

Same commands, options, statistics and default output names as /root/reference/bootstrapper/refine.py (`size_filter`
:176-213, `outlier_filter` :131-168, `z_filter` :221-257, `remap` :272-307; `morph` needs fastmorph [EXT] and is not part
of this engine).  Structure: the label volume is scanned once on the device into an `ObjectTable` (ids, voxel counts,
first / last section: `bsmi_label_table_u64` tile by tile), a filter is a rule that turns the table into a rejection
mask (`FILTERS`), and `run_filter` writes the volume with the rejected ids zeroed through `bsmi_lut_relabel`.
"""
import click
import numpy as np

from .zarr_io import open_ds, prepare_ds


def derived_dataset(in_array, suffix):
    """`<store>.zarr/<dataset>` -> `<store>.zarr/<dataset>_<suffix>`, the reference's default output name; an input that
    does not sit directly inside exactly one .zarr container has no such default."""
    marker = ".zarr/"
    at = in_array.find(marker)
    if at < 0 or marker in in_array[at + len(marker):]:
        raise click.ClickException(f"no default output name for {in_array!r}: give --out_array")
    return f"{in_array}_{suffix}"


def _empty_copy(in_ds, out_array):
    """an output dataset with the input's geometry, metadata, chunking and compressor"""
    keep = dict(shape=in_ds.shape, dtype=in_ds.dtype, chunk_shape=in_ds.chunks, offset=in_ds.offset, voxel_size=in_ds.voxel_size,
                axis_names=in_ds.axis_names, units=in_ds.units, compressor=in_ds.meta.get("compressor"))
    return prepare_ds(out_array, **keep)


def _tile_rows(in_ds):
    """rows (y) of a tile: whole rows of x (what the dataset writers want: `post.watershed._LayerWriter`), whole chunks, and a
    section of fewer than 2^20 voxels (bsmi_seg_create)"""
    ny, nx = in_ds.shape[1], in_ds.shape[2]
    cy = int(in_ds.chunks[1])
    most = max(1, ((1 << 20) - 1) // max(1, nx))
    return ny if ny <= most else max(cy if cy <= most else most, most // cy * cy)


def _tiles(in_ds, tile=None):
    nz, ny, nx = in_ds.shape
    cz = int(in_ds.chunks[0])
    rows = _tile_rows(in_ds) if tile is None else tile
    for iz in range(0, nz, cz):
        for iy in range(0, ny, rows):
            yield iz, (slice(iz, min(iz + cz, nz)), slice(iy, min(iy + rows, ny)), slice(0, nx))


class _Device:
    def __init__(self, in_ds, device=0):
        import torch
        from .post.engine import SegEngine
        self.torch = torch
        self.dev = torch.device("cuda", device)
        cz = int(in_ds.chunks[0])
        if in_ds.shape[2] >= (1 << 20):
            raise NotImplementedError("rows of 2^20 voxels or more")
        self.engine = SegEngine((min(cz, in_ds.shape[0]), min(_tile_rows(in_ds), in_ds.shape[1]), in_ds.shape[2]), device)
        self.tile_edge = None

        # tiles read for the object table stay on the device for the pass that writes the filtered copy (a rule is evaluated
        # between two passes over the same volume: the second read and decode of a dataset that fits is saved)
        self.kept, self.kept_bytes, self.budget = {}, 0, 10 << 30

    def upload(self, block):
        return self.torch.from_numpy(np.ascontiguousarray(block).astype(np.uint64).view(np.int64)).to(self.dev)

    def tile(self, in_ds, sl, keep=False):
        key = tuple((s.start, s.stop) for s in sl)
        t = self.kept.get(key)
        if t is None:
            t = self.upload(in_ds[sl])
            if keep and self.kept_bytes + t.numel() * 8 <= self.budget:
                self.kept[key] = t
                self.kept_bytes += t.numel() * 8
        return t


def label_table(in_ds, device=0, holder=None):
    """-> (ids ascending u64, sizes, zmin, zmax) over the whole volume (refine.py:98-109, 228-250)."""
    d = holder or _Device(in_ds, device)
    parts = []
    for iz, sl in _tiles(in_ds, d.tile_edge):
        parts.append(d.engine.label_table(d.tile(in_ds, sl, keep=holder is not None), iz))
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


def _apply_mapping(in_ds, out_array, keys, vals, device=0, holder=None):
    """out = in with keys[k] -> vals[k] (ids not listed stay), tile by tile through bsmi_lut_relabel."""
    from .post.engine import lut_relabel
    d = holder or _Device(in_ds, device)
    out_ds = _empty_copy(in_ds, out_array)
    order = np.argsort(keys, kind="stable")
    k = d.torch.from_numpy(np.asarray(keys, np.uint64)[order].view(np.int64)).to(d.dev)
    v = d.torch.from_numpy(np.asarray(vals, np.uint64)[order].view(np.int64)).to(d.dev)
    if in_ds.dtype == np.uint64:
        # write-behind through the dataset writer of `bs segment`: Blosc frames made on the device where the dataset allows
        # (csrc/blosc_dev.hip), files written on a pool while the next tiles are relabelled
        from .post.watershed import _LayerWriter
        writer = _LayerWriter(d.dev, int(in_ds.chunks[0]))
        try:
            for _, sl in _tiles(in_ds, d.tile_edge):
                writer.submit(out_ds, lut_relabel(d.tile(in_ds, sl), k, v), sl[0].start, sl[1].start)
            writer.drain()
        finally:
            writer.close()
        return out_ds
    for _, sl in _tiles(in_ds, d.tile_edge):
        lab = d.tile(in_ds, sl)
        out_ds[sl] = lut_relabel(lab, k, v).cpu().numpy().view(np.uint64).astype(in_ds.dtype)
    return out_ds


class ObjectTable:
    """Per-object statistics of a label volume, gathered on the device: ids ascending, voxel counts, first / last section."""

    def __init__(self, in_array, device=0):
        self.path = in_array
        self.ds = open_ds(in_array)
        self.holder = _Device(self.ds, device)
        self.ids, self.sizes, self.zmin, self.zmax = label_table(self.ds, device, self.holder)
        if self.ids.size == 0:
            raise click.ClickException(f"{in_array} holds no labelled voxels")

    @property
    def z_extent(self):
        return self.zmax - self.zmin + 1


def _by_size(t, min_size=0, max_size=None, **_):
    """objects below min_size (when set) or above max_size (when set)"""
    drop = np.zeros(t.ids.size, dtype=bool)
    if min_size > 0:
        drop |= t.sizes < min_size
    if max_size:
        drop |= t.sizes > max_size
    yield f"{t.ids.size} objects, {int(t.sizes.min())} to {int(t.sizes.max())} voxels (median {int(np.median(t.sizes))})"
    yield f"keeping sizes in [{min_size}, {max_size}]: {int(drop.sum())} objects go"
    return drop


def _by_outlier(t, num_std=3.0, min_size=0, **_):
    """objects whose size is further than num_std standard deviations from the mean; both moments are taken over the
    objects of at least min_size voxels (population standard deviation), the cut applies to all"""
    basis = t.sizes[t.sizes >= min_size]
    if basis.size == 0:
        raise click.ClickException(f"no object reaches min_size = {min_size}: no statistics to cut by")
    centre, spread = float(basis.mean()), float(basis.std())
    low, high = centre - num_std * spread, centre + num_std * spread
    drop = (t.sizes < low) | (t.sizes > high)
    q = np.percentile(basis, [50, 90, 99, 99.9])
    yield f"{t.ids.size} objects, {basis.size} of them >= {min_size} voxels; quantiles 50/90/99/99.9 %: " + " / ".join(f"{v:.0f}" for v in q)
    yield f"mean {centre:.1f}, std {spread:.1f}: sizes outside [{low:.1f}, {high:.1f}] go ({int(drop.sum())} objects)"
    return drop


def _by_z_extent(t, min_z=1, **_):
    """objects that span min_z sections or fewer"""
    drop = t.z_extent <= min_z
    yield f"{t.ids.size} objects; {int(drop.sum())} span {min_z} section(s) or fewer and go"
    return drop


# filter name -> (rule, suffix of the default output dataset)
FILTERS = {"size": (_by_size, "size_filtered"), "outlier": (_by_outlier, "outlier_filtered"), "z": (_by_z_extent, "z_filtered")}


def run_filter(kind, in_array, out_array=None, dry_run=False, device=0, **options):
    """Evaluate one rule on the object table and write the volume with the rejected objects set to 0 (refine.py:131-257).
    -> the dataset written, None on a dry run."""
    rule, suffix = FILTERS[kind]
    table = ObjectTable(in_array, device)
    steps = rule(table, **options)
    try:
        while True:
            print(next(steps))
    except StopIteration as done:
        drop = done.value
    if dry_run:
        print("dry run: no dataset written")
        return None
    target = out_array or derived_dataset(in_array, suffix)
    print(f"-> {target}")
    gone = table.ids[drop]
    _apply_mapping(table.ds, target, gone, np.zeros(gone.size, np.uint64), device, table.holder)
    return target


def size_filter(in_array, out_array=None, min_size=0, max_size=None, dry_run=False, device=0):
    return run_filter("size", in_array, out_array, dry_run, device, min_size=min_size, max_size=max_size)


def outlier_filter(in_array, out_array=None, num_std=3.0, min_size=0, dry_run=False, device=0):
    return run_filter("outlier", in_array, out_array, dry_run, device, num_std=num_std, min_size=min_size)


def z_filter(in_array, out_array=None, min_z=1, dry_run=False, device=0):
    return run_filter("z", in_array, out_array, dry_run, device, min_z=min_z)


def _id_list(text):
    return [int(tok) for tok in text.replace(" ", "").split(",") if tok]


def remap(in_array, out_array=None, remove_ids=None, merge_ids=(), device=0):
    """ids of --remove_ids become 0; every id of a --merge_ids group becomes the group's first id (refine.py:272-307)"""
    table = {}
    for group in merge_ids:
        members = _id_list(group)
        table.update({m: members[0] for m in members})
    zeroed = _id_list(remove_ids) if remove_ids else []
    both = sorted(set(zeroed) & set(table))
    if both:
        raise click.ClickException(f"{both}: an id cannot be removed and merged at once")
    table.update({z: 0 for z in zeroed})
    if not table:
        raise click.ClickException("give --remove_ids and / or --merge_ids")
    print(f"{len(table)} ids change: {table}")
    target = out_array or derived_dataset(in_array, "remapped")
    print(f"-> {target}")
    src = np.fromiter(sorted(table), dtype=np.uint64, count=len(table))
    _apply_mapping(open_ds(in_array), target, src, np.array([table[int(k)] for k in src], dtype=np.uint64), device)
    return target


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
