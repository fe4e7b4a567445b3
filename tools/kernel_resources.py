"""Per-kernel resources of the gfx950 code objects inside libbsmi.so: registers, LDS, and the scratch (private) segment.
`python tools/kernel_resources.py [--scratch]` lists them; tests/test_kernel_resources.py holds the product path to
"no scratch segment" (the kernel two overlapping forward passes corrupted was the one with scratch)."""
import os, struct, sys
import msgpack

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _section(elf, name):
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", elf, 0x3A)
    heads = [struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize) for i in range(shnum)]
    strtab = heads[shstrndx]
    for h in heads:
        end = elf.index(b"\0", strtab[4] + h[0])
        if elf[strtab[4] + h[0]:end] == name:
            return elf[h[4]:h[4] + h[5]]
    return None


def _code_objects(blob):
    """gfx950 ELF images of every offload bundle in a .hip_fatbin section (one bundle per translation unit)."""
    at = blob.find(MAGIC)
    while at >= 0:
        n, = struct.unpack_from("<Q", blob, at + len(MAGIC))
        p = at + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                yield blob[at + off:at + off + size]
        at = blob.find(MAGIC, at + 1)


def kernels(so_path):
    """[{name, vgprs, agprs, sgprs, lds, scratch}] of every kernel in the library."""
    so = open(so_path, "rb").read()
    fat = _section(so, b".hip_fatbin")
    if fat is None:
        raise RuntimeError(f"{so_path}: no .hip_fatbin section")
    out = []
    for elf in _code_objects(fat):
        note = _section(elf, b".note")
        p = 0
        while note is not None and p + 12 <= len(note):
            namesz, descsz, typ = struct.unpack_from("<III", note, p)
            p += 12
            name = note[p:p + namesz]; p += (namesz + 3) & ~3
            desc = note[p:p + descsz]; p += (descsz + 3) & ~3
            if typ != 32 or not name.startswith(b"AMDGPU"):
                continue
            meta = msgpack.unpackb(desc, raw=False, strict_map_key=False)
            for k in meta.get("amdhsa.kernels", []):
                out.append({"name": k[".name"], "vgprs": k.get(".vgpr_count", 0), "agprs": k.get(".agpr_count", 0),
                            "sgprs": k.get(".sgpr_count", 0), "lds": k.get(".group_segment_fixed_size", 0),
                            "scratch": k.get(".private_segment_fixed_size", 0),
                            "dynamic_stack": bool(k.get(".uses_dynamic_stack", False))})
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ks = kernels(os.path.join(here, "bootstrapper_amd", "libbsmi.so"))
    only = "--scratch" in sys.argv
    for k in sorted(ks, key=lambda k: -k["scratch"]):
        if only and not (k["scratch"] or k["dynamic_stack"]): continue
        print(f'{k["scratch"]:6d} B/lane scratch  {k["vgprs"]:3d}+{k["agprs"]:3d} vgpr  {k["lds"]:6d} B lds  {k["name"][:150]}')
    print(f"{len(ks)} kernels, {sum(1 for k in ks if k['scratch'])} with a scratch segment")
