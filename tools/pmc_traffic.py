"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into HBM-side bytes per conv launch.

usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [conv steps in the run]

A conv step of the plan is one launch of the implicit-GEMM kernel or, in its Winograd form (csrc/wino.hip), the input
transform + the batched GEMM launch + (the residual launch) + the output transform: all of them are counted, and the
per-step figure divides by the number of conv steps of the profiled run (tools/probe_unet.py: 3 forwards x 13 steps),
given as the fourth argument; without it the launches of the conv kernels are counted as before round 3.

Corrections of /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE tallies the 128-byte requests of wide (16 B per lane) reads at 64 bytes, so it is
doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.  Infinity-Cache hits are counted, i.e. this is
traffic on the memory side of L2, not DRAM traffic proper.
"""
import collections
import csv
import json
import sys


def load(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        agg[name][0] += 1
        agg[name][1] += float(r["Counter_Value"])
    return agg


def main():
    fetch, write, out = sys.argv[1:4]
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    f, w = load(fetch, "FETCH_SIZE"), load(write, "WRITE_SIZE")
    kernels = {}
    tot_l = tot_f = tot_w = 0
    for name in sorted(set(f) | set(w)):
        if "conv_" not in name and "wino" not in name and "first_pass" not in name:
            continue
        n = f.get(name, [0, 0])[0] or w.get(name, [0, 0])[0]
        fb = f.get(name, [0, 0.0])[1] * 1024 * 2
        wb = w.get(name, [0, 0.0])[1] * 1024
        kernels[name] = {"launches": n, "fetch_bytes_per_launch": fb / max(n, 1), "write_bytes_per_launch": wb / max(n, 1)}
        if "fixup" not in name:
            tot_l += n
        tot_f += fb
        tot_w += wb
    if steps:
        tot_l = steps
    summary = {"conv_launches": tot_l, "traffic_bytes_per_conv_launch": (tot_f + tot_w) / max(tot_l, 1),
               "fetch_bytes_per_conv_launch": tot_f / max(tot_l, 1), "write_bytes_per_conv_launch": tot_w / max(tot_l, 1),
               "corrections": "KiB -> bytes; FETCH_SIZE x2 (gfx950 wide reads); Infinity-Cache hits included", "kernels": kernels}
    json.dump(summary, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in summary.items() if k != "kernels"}))


if __name__ == "__main__":
    main()
