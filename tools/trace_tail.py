"""Dev tool: what happens between the last predicted block and the end of the timed job, from a rocprofv3 --kernel-trace
CSV of `bench.py --steps N --warmup W --no-modes --no-train --no-drivers --no-cpu-baseline`.
usage: trace_tail.py <kernel_trace.csv> <warmup + steps>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
nheads = int(sys.argv[2])
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows)
heads = [e for e in ev if "head_kernel" in e[2]]
t0 = heads[nheads - 1][1]
relabels = [e for e in ev if "lut_relabel" in e[2] and e[0] > t0]
t1 = relabels[0][1] if "multi" in relabels[0][2] else relabels[2][1]
print(f"tail: last predicted block -> end of the relabel = {(t1 - t0) / 1e6:.1f} ms")
win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
GROUPS = [("seeds", ("ws_seeds",)), ("flood", ("ws_flood",)), ("frag post", ("frag_", "cc26_", "crop_u64", "label_stats", "ws_offsets")),
          ("rag scans", ("rag_ids", "agg_edges", "agg_compact", "rag_", "rocprim", "seg_clear")), ("merge loop", ("rag_merge_kernel",)),
          ("fills / copies", ("fillBuffer", "copyBuffer", "elementwise", "at::")), ("relabel", ("lut_relabel",))]
def group(name):
    if "rag_merge_kernel" in name:
        return "merge loop"
    for g, keys in GROUPS:
        if any(k in name for k in keys):
            return g
    return "other"
acc = defaultdict(lambda: [0, 0.0, 1e30, 0])
for s, e, n, q in win:
    a = acc[group(n)]
    a[0] += 1; a[1] += (e - s) / 1e6; a[2] = min(a[2], s); a[3] = max(a[3], e)
print(f"{'group':16s} {'launches':>8s} {'kernel ms':>10s} {'first start':>12s} {'last end':>10s}   (ms after the last predicted block)")
for g, (c, t, a, b) in sorted(acc.items(), key=lambda kv: kv[1][2]):
    print(f"{g:16s} {c:8d} {t:10.1f} {(a - t0) / 1e6:12.1f} {(b - t0) / 1e6:10.1f}")
perq = defaultdict(list)
for s, e, n, q in win:
    perq[q].append((s, e, n))
print("per queue: kernels, busy ms, first start, last end, longest idle gap (ms) and what followed it")
for q, es in sorted(perq.items(), key=lambda kv: kv[1][0][0]):
    busy = sum(e - s for s, e, _ in es) / 1e6
    gaps = [(es[i + 1][0] - es[i][1], es[i + 1][2]) for i in range(len(es) - 1)]
    g, after = max(gaps) if gaps else (0, "")
    print(f"  queue {q:>4s}: {len(es):5d} {busy:8.1f} {(es[0][0] - t0) / 1e6:8.1f} {(es[-1][1] - t0) / 1e6:8.1f}   gap {g / 1e6:6.1f} before {after.split('(')[0][-40:]}")
# how many flood / merge kernels run at a time
for key in ("ws_flood", "rag_merge_kernel"):
    pts = []
    for s, e, n, q in win:
        if key in n:
            pts += [(s, 1), (e, -1)]
    pts.sort()
    cur = peak = 0
    for _, d in pts:
        cur += d; peak = max(peak, cur)
    print(f"{key}: {len(pts) // 2} launches, at most {peak} at a time")
