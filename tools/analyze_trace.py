"""Timeline of the predict stream from a rocprofv3 --kernel-trace CSV of bench.py: per block, time inside U-Net
kernels, gaps between consecutive U-Net kernels, and what ran in the gaps.  Usage: analyze_trace.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
print("columns:", list(rows[0].keys()))
UNET = ("conv_igemm", "conv_rh", "conv_box", "conv_fixup", "first_pass", "maxpool", "upsample", "head_kernel", "input_prep", "extract_block")
ev = []
for r in rows:
    name = r["Kernel_Name"]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", ""), r.get("Stream_Id", "")))
ev.sort()
unet = [e for e in ev if any(k in e[2] for k in UNET)]
heads = [i for i, e in enumerate(unet) if "head_kernel" in e[2]]
print("unet kernels", len(unet), "blocks", len(heads))
# steady-state window: blocks 4 .. last-2
lo, hi = heads[3] + 1, heads[-3] + 1
win = unet[lo:hi]
nblk = len([e for e in win if "head_kernel" in e[2]])
busy = sum(e[1] - e[0] for e in win)
span = win[-1][1] - win[0][0]
gaps = [(win[i + 1][0] - win[i][1], win[i][2][:40], win[i + 1][2][:40]) for i in range(len(win) - 1)]
gap_total = sum(max(g[0], 0) for g in gaps)
print(f"blocks {nblk}: span {span / nblk / 1e6:.2f} ms/block, in U-Net kernels {busy / nblk / 1e6:.2f}, gaps {gap_total / nblk / 1e6:.2f}")
by = defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    key = (a.split("<")[0].split("(")[0][-28:], b.split("<")[0].split("(")[0][-28:])
    by[key][0] += max(g, 0)
    by[key][1] += 1
for k, (t, n) in sorted(by.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  gap {k[0]:>28s} -> {k[1]:<28s} {t / nblk / 1e3:8.1f} us/block  ({n / nblk:.1f} per block, {t / max(n, 1) / 1e3:.1f} us each)")
dur = defaultdict(lambda: [0, 0])
for s, e, name, *_ in win:
    k = name.split("(")[0][-60:]
    dur[k][0] += e - s
    dur[k][1] += 1
for k, (t, n) in sorted(dur.items(), key=lambda kv: -kv[1][0])[:12]:
    print(f"  kern {k:60s} {t / nblk / 1e6:7.3f} ms/block ({n / nblk:.1f} launches)")
